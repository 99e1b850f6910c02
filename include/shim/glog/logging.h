// include/shim/glog/logging.h — the slice of glog the reference's sources use, for boxes without
// glog (this image).  Call sites served: LOG / VLOG / CHECK* / CHECK_NOTNULL throughout
// src/dqn_main.cpp, src/dqn.cpp, src/hfo_game.cpp; google::InitGoogleLogging,
// InstallFailureSignalHandler, LogToStderr, SetLogDestination, GLOG_* and
// fLI::FLAGS_logbuflevel at src/dqn_main.cpp:394-409.  Error convention kept: LOG(FATAL) and a
// failed CHECK print the message and abort() (no exceptions, no status codes).
#ifndef DQNHIP_SHIM_GLOG_LOGGING_H_
#define DQNHIP_SHIM_GLOG_LOGGING_H_

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <mutex>
#include <sstream>
#include <string>
#include <unistd.h>      // the real glog headers bring it in; src/hfo_game.cpp:37 calls sleep()

namespace fLI {
inline int FLAGS_logbuflevel = 0;     // set by src/dqn_main.cpp:396
inline int FLAGS_v = 0;               // VLOG threshold (env GLOG_v)
}  // namespace fLI

namespace google {

typedef int LogSeverity;
const LogSeverity GLOG_INFO = 0, GLOG_WARNING = 1, GLOG_ERROR = 2, GLOG_FATAL = 3;

namespace shim {
inline std::mutex& mu() { static std::mutex m; return m; }
inline FILE*& sink(int sev) { static FILE* f[4] = {nullptr, nullptr, nullptr, nullptr}; return f[sev]; }
inline bool& to_stderr() { static bool b = false; return b; }
inline int vlevel() {
  static int v = [] { const char* e = std::getenv("GLOG_v"); return e ? std::atoi(e) : 0; }();
  return v > fLI::FLAGS_v ? v : fLI::FLAGS_v;
}
}  // namespace shim

inline void InitGoogleLogging(const char*) {}
inline void InstallFailureSignalHandler() {}
inline void LogToStderr() { shim::to_stderr() = true; }
// glog appends "<date>-<time>.<pid>" to the base; here one file per severity: <base>log
inline void SetLogDestination(LogSeverity sev, const char* base) {
  if (sev < 0 || sev > 3) return;
  std::lock_guard<std::mutex> lk(shim::mu());
  if (shim::sink(sev)) { std::fclose(shim::sink(sev)); shim::sink(sev) = nullptr; }
  if (base && *base) shim::sink(sev) = std::fopen((std::string(base) + "log").c_str(), "a");
}

class LogMessage {
 public:
  LogMessage(const char* file, int line, LogSeverity sev) : sev_(sev) {
    const char* slash = std::strrchr(file, '/');
    s_ << "IWEF"[sev] << ' ' << (slash ? slash + 1 : file) << ':' << line << "] ";
  }
  ~LogMessage() {
    s_ << '\n';
    const std::string m = s_.str();
    {
      std::lock_guard<std::mutex> lk(shim::mu());
      // a message goes to its own severity's file and to every lower one, as in glog
      bool filed = false;
      for (int s = sev_; s >= 0; --s) if (shim::sink(s)) { std::fputs(m.c_str(), shim::sink(s)); std::fflush(shim::sink(s)); filed = true; }
      if (!filed || shim::to_stderr() || sev_ >= GLOG_ERROR) { std::fputs(m.c_str(), stderr); std::fflush(stderr); }
    }
    if (sev_ == GLOG_FATAL) std::abort();
  }
  std::ostream& stream() { return s_; }

 private:
  std::ostringstream s_;
  LogSeverity sev_;
};

struct LogMessageVoidify { void operator&(std::ostream&) {} };

template <typename T>
T&& CheckNotNull(const char* file, int line, const char* expr, T&& t) {
  if (t == nullptr) LogMessage(file, line, GLOG_FATAL).stream() << "Check failed: '" << expr << "' Must be non NULL";
  return std::forward<T>(t);
}

}  // namespace google

#define LOG(severity) ::google::LogMessage(__FILE__, __LINE__, ::google::GLOG_##severity).stream()
#define VLOG_IS_ON(n) ((n) <= ::google::shim::vlevel())
#define VLOG(n) !VLOG_IS_ON(n) ? (void)0 : ::google::LogMessageVoidify() & LOG(INFO)
#define CHECK(cond) (cond) ? (void)0 : ::google::LogMessageVoidify() & LOG(FATAL) << "Check failed: " #cond " "
#define DQNHIP_SHIM_CHECK_OP(a, op, b)                                                                  \
  ((a)op(b)) ? (void)0 : ::google::LogMessageVoidify() & LOG(FATAL) << "Check failed: " #a " " #op " " #b \
                                                                    << " (" << (a) << " vs. " << (b) << ") "
#define CHECK_EQ(a, b) DQNHIP_SHIM_CHECK_OP(a, ==, b)
#define CHECK_NE(a, b) DQNHIP_SHIM_CHECK_OP(a, !=, b)
#define CHECK_LT(a, b) DQNHIP_SHIM_CHECK_OP(a, <, b)
#define CHECK_LE(a, b) DQNHIP_SHIM_CHECK_OP(a, <=, b)
#define CHECK_GT(a, b) DQNHIP_SHIM_CHECK_OP(a, >, b)
#define CHECK_GE(a, b) DQNHIP_SHIM_CHECK_OP(a, >=, b)
#define CHECK_NOTNULL(p) ::google::CheckNotNull(__FILE__, __LINE__, #p, (p))

#endif  // DQNHIP_SHIM_GLOG_LOGGING_H_
