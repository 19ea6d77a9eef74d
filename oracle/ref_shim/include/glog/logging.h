// oracle/ref_shim: the glog macros the reference's registration path uses (TEST INFRASTRUCTURE ONLY).
// LOG / DLOG lines are collected in a small per-thread ring (ref_api.cpp reads the "num iter=" line to report
// the executed iteration count); CHECK failures throw ref_shim::CheckFailure instead of aborting the test process.
#pragma once
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace ref_shim {
struct CheckFailure : std::runtime_error { using std::runtime_error::runtime_error; };
inline std::vector<std::string>& log_lines() { static thread_local std::vector<std::string> v; return v; }
struct LogMessage {
    std::ostringstream os; bool fatal;
    explicit LogMessage(bool f = false) : fatal(f) {}
    ~LogMessage() noexcept(false) {
        auto& v = log_lines();
        if (v.size() > 256) v.erase(v.begin(), v.begin() + 128);
        v.push_back(os.str());
        if (fatal) throw CheckFailure(os.str());
    }
    std::ostream& stream() { return os; }
};
struct Voidify { void operator&(std::ostream&) {} };
template <class T> T check_notnull(T&& p, const char* what) { if (p == nullptr) throw CheckFailure(what); return std::forward<T>(p); }
}  // namespace ref_shim

#define LOG(sev) REF_SHIM_LOG_##sev
#define DLOG(sev) REF_SHIM_LOG_##sev
#define REF_SHIM_LOG_INFO ::ref_shim::LogMessage().stream()
#define REF_SHIM_LOG_WARNING ::ref_shim::LogMessage().stream()
#define REF_SHIM_LOG_ERROR ::ref_shim::LogMessage().stream()
#define REF_SHIM_LOG_FATAL ::ref_shim::LogMessage(true).stream()
#define REF_SHIM_CHECK(cond, text) (cond) ? (void)0 : ::ref_shim::Voidify() & ::ref_shim::LogMessage(true).stream() << "Check failed: " text " "
#define CHECK(c) REF_SHIM_CHECK((c), #c)
#define CHECK_EQ(a, b) REF_SHIM_CHECK((a) == (b), #a " == " #b)
#define CHECK_NE(a, b) REF_SHIM_CHECK((a) != (b), #a " != " #b)
#define CHECK_GT(a, b) REF_SHIM_CHECK((a) > (b), #a " > " #b)
#define CHECK_GE(a, b) REF_SHIM_CHECK((a) >= (b), #a " >= " #b)
#define CHECK_LT(a, b) REF_SHIM_CHECK((a) < (b), #a " < " #b)
#define CHECK_LE(a, b) REF_SHIM_CHECK((a) <= (b), #a " <= " #b)
#define CHECK_NOTNULL(p) ::ref_shim::check_notnull((p), "Check failed: '" #p "' Must be non NULL")
