// The C-ABI's status codes ARE gr::work::Status (include/gr4hip.h): checked against the reference's own core/include/gnuradio-4.0/WorkStatus.hpp, included from
// /root/reference where it lies, unmodified (std-only header).  Container-only, compile-time only.
#include <gnuradio-4.0/WorkStatus.hpp> // the reference's file

#include "../../../include/gr4hip.h"

static_assert(static_cast<int>(gr::work::Status::OK) == GR4HIP_OK);
static_assert(static_cast<int>(gr::work::Status::DONE) == GR4HIP_DONE);
static_assert(static_cast<int>(gr::work::Status::INSUFFICIENT_INPUT_ITEMS) == GR4HIP_INSUFFICIENT_INPUT);
static_assert(static_cast<int>(gr::work::Status::INSUFFICIENT_OUTPUT_ITEMS) == GR4HIP_INSUFFICIENT_OUTPUT);
static_assert(static_cast<int>(gr::work::Status::ERROR) == GR4HIP_ERROR);
int main() { return 0; }
