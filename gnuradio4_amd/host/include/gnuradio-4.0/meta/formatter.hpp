// forwarding header: lets block sources written for the reference's include tree compile unchanged on this host layer (gr::exception and the message types live in gr4/compat.hpp / gr4/core.hpp)
#pragma once
#include "../../gr4/compat.hpp"
