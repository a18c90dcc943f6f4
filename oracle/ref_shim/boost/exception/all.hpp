// TEST INFRASTRUCTURE (oracle/ref_shim): the three Boost.Exception names regenie mentions
// (src/Regenie.cpp:83-90, src/Data.cpp:2539) on top of <exception>.
#ifndef RG_SHIM_BOOST_EXCEPTION_ALL_HPP
#define RG_SHIM_BOOST_EXCEPTION_ALL_HPP
#include <exception>
#include <string>
namespace boost {
struct exception { virtual ~exception() {} };
inline std::string diagnostic_information(const exception&) { return "boost::exception (shim)"; }
inline std::string current_exception_diagnostic_information() {
  try { throw; }
  catch (const std::exception& e) { return e.what(); }
  catch (const std::string& s) { return s; }
  catch (const char* s) { return s; }
  catch (...) { return "unknown exception"; }
}
}
#endif
