// forwarding header: lets block sources written for the reference's include tree (#include <gnuradio-4.0/annotated.hpp>) compile unchanged on this host layer
#pragma once
#include "../gr4/compat.hpp"
