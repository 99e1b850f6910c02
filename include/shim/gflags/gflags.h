// include/shim/gflags/gflags.h — the slice of gflags the reference's sources use, for boxes
// without gflags (this image).  NOT the product: a build that has the real gflags simply
// leaves include/shim off its include path.
//
// Call sites served (reference): DEFINE_{bool,int32,double,string} src/dqn_main.cpp:18-59,
// src/dqn.cpp:21-31, src/hfo_game.cpp:8-20; gflags::SetUsageMessage / SetVersionString /
// ParseCommandLineFlags / ProgramUsage src/dqn_main.cpp:389-402.  As in gflags proper, a flag
// lives in namespace fLB / fLI / fLD / fLS (src/dqn_main.cpp:396 names fLI::FLAGS_logbuflevel)
// and every DEFINE in any translation unit registers in ONE process-wide table, which is what
// lets the driver's ParseCommandLineFlags set the learner flags defined by the library.
#ifndef DQNHIP_SHIM_GFLAGS_H_
#define DQNHIP_SHIM_GFLAGS_H_

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace gflags {
namespace shim {

enum Kind { K_BOOL, K_INT32, K_DOUBLE, K_STRING };
struct Flag { Kind kind; void* ptr; std::string help; };

inline std::map<std::string, Flag>& table() { static std::map<std::string, Flag> t; return t; }
inline std::string& usage() { static std::string u; return u; }
inline std::string& version() { static std::string v; return v; }

struct Registrar {
  Registrar(const char* name, Kind kind, void* ptr, const char* help) { table()[name] = Flag{kind, ptr, help}; }
};

inline bool assign(const Flag& f, const std::string& name, const std::string& value) {
  char* end = nullptr;
  switch (f.kind) {
    case K_BOOL:
      if (value == "true" || value == "1" || value == "t" || value == "yes" || value == "y") { *static_cast<bool*>(f.ptr) = true; return true; }
      if (value == "false" || value == "0" || value == "f" || value == "no" || value == "n") { *static_cast<bool*>(f.ptr) = false; return true; }
      break;
    case K_INT32: { const long v = std::strtol(value.c_str(), &end, 0); if (end && *end == 0 && !value.empty()) { *static_cast<int32_t*>(f.ptr) = (int32_t)v; return true; } break; }
    case K_DOUBLE: { const double v = std::strtod(value.c_str(), &end); if (end && *end == 0 && !value.empty()) { *static_cast<double*>(f.ptr) = v; return true; } break; }
    case K_STRING: *static_cast<std::string*>(f.ptr) = value; return true;
  }
  std::fprintf(stderr, "ERROR: illegal value '%s' specified for flag '%s'\n", value.c_str(), name.c_str());
  return false;
}

}  // namespace shim

inline void SetUsageMessage(const std::string& u) { shim::usage() = u; }
inline void SetVersionString(const std::string& v) { shim::version() = v; }
inline const char* ProgramUsage() { return shim::usage().c_str(); }

// -flag=value, --flag=value, -flag value, --flag value; booleans also -flag / -noflag.
// "--" ends flag parsing.  Unknown flags are fatal, as in gflags.
inline uint32_t ParseCommandLineFlags(int* argc, char*** argv, bool remove_flags) {
  std::vector<char*> rest;
  rest.push_back((*argv)[0]);
  bool ok = true, flags_done = false;
  for (int i = 1; i < *argc; ++i) {
    char* arg = (*argv)[i];
    if (flags_done || arg[0] != '-' || arg[1] == 0) { rest.push_back(arg); continue; }
    if (!std::strcmp(arg, "--")) { flags_done = true; continue; }
    std::string body(arg + (arg[1] == '-' ? 2 : 1));
    std::string name = body, value;
    bool has_value = false;
    const size_t eq = body.find('=');
    if (eq != std::string::npos) { name = body.substr(0, eq); value = body.substr(eq + 1); has_value = true; }
    auto& t = shim::table();
    auto it = t.find(name);
    if (it == t.end() && !has_value && name.rfind("no", 0) == 0) {           // -noflag
      auto nt = t.find(name.substr(2));
      if (nt != t.end() && nt->second.kind == shim::K_BOOL) { *static_cast<bool*>(nt->second.ptr) = false; continue; }
    }
    if (it == t.end()) { std::fprintf(stderr, "ERROR: unknown command line flag '%s'\n", name.c_str()); ok = false; continue; }
    if (!has_value) {
      if (it->second.kind == shim::K_BOOL) { *static_cast<bool*>(it->second.ptr) = true; continue; }
      if (i + 1 >= *argc) { std::fprintf(stderr, "ERROR: flag '%s' is missing its argument\n", name.c_str()); ok = false; continue; }
      value = (*argv)[++i];
    }
    ok = shim::assign(it->second, name, value) && ok;
  }
  if (!ok) std::exit(1);
  if (remove_flags) {
    for (size_t k = 0; k < rest.size(); ++k) (*argv)[k] = rest[k];
    *argc = (int)rest.size();
    return 1;
  }
  return (uint32_t)*argc;
}

}  // namespace gflags

#define DQNHIP_SHIM_DEFINE(ns, ctype, kind, name, value, help)                                       \
  namespace ns {                                                                                     \
  ctype FLAGS_##name = value;                                                                        \
  static ::gflags::shim::Registrar shim_reg_##name(#name, ::gflags::shim::kind, &FLAGS_##name, help); \
  }                                                                                                  \
  using ns::FLAGS_##name
#define DEFINE_bool(name, value, help) DQNHIP_SHIM_DEFINE(fLB, bool, K_BOOL, name, value, help)
#define DEFINE_int32(name, value, help) DQNHIP_SHIM_DEFINE(fLI, ::int32_t, K_INT32, name, value, help)
#define DEFINE_double(name, value, help) DQNHIP_SHIM_DEFINE(fLD, double, K_DOUBLE, name, value, help)
#define DEFINE_string(name, value, help) DQNHIP_SHIM_DEFINE(fLS, ::std::string, K_STRING, name, value, help)

#define DQNHIP_SHIM_DECLARE(ns, ctype, name) \
  namespace ns { extern ctype FLAGS_##name; } \
  using ns::FLAGS_##name
#define DECLARE_bool(name) DQNHIP_SHIM_DECLARE(fLB, bool, name)
#define DECLARE_int32(name) DQNHIP_SHIM_DECLARE(fLI, ::int32_t, name)
#define DECLARE_double(name) DQNHIP_SHIM_DECLARE(fLD, double, name)
#define DECLARE_string(name) DQNHIP_SHIM_DECLARE(fLS, ::std::string, name)

#endif  // DQNHIP_SHIM_GFLAGS_H_
