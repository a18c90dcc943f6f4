// TEST INFRASTRUCTURE (oracle/ref_shim): forwards to the one shim header.
#include <boost/math/distributions.hpp>
