// TEST INFRASTRUCTURE (oracle/ref_shim): boost::filesystem as used by regenie (path, absolute,
// extension: src/Data.cpp:1153-1170, src/Files.cpp:144) mapped onto C++17 <filesystem>.
#ifndef RG_SHIM_BOOST_FILESYSTEM_HPP
#define RG_SHIM_BOOST_FILESYSTEM_HPP
#include <filesystem>
#include <string>
namespace boost { namespace filesystem {
using std::filesystem::path;
using std::filesystem::absolute;
using std::filesystem::exists;
inline std::string extension(const std::string& f) { return std::filesystem::path(f).extension().string(); }
}}
#endif
