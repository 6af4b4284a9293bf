// A minimal stand-in for googletest (not installed in this image; no network): TEST + the ASSERT_* forms the reference's GPU acceptance
// tests use (test/evaluator_cuda.cu, encryptor_cuda.cu, ckks_cuda.cu).  TEST INFRASTRUCTURE ONLY.  A failed assertion throws (the
// reference's tests assert inside lambdas and loops), the runner catches per test, prints gtest-style lines and returns non-zero on failure.
// Run with an optional filter argument: a substring of "Suite.Name" (or --gtest_filter=<substring>).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <chrono>
#include <functional>
#include <iostream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace gtest_shim {
struct Failure : std::runtime_error { using std::runtime_error::runtime_error; };
struct Case { const char *suite, *name; void (*fn)(); };
inline std::vector<Case> &registry() { static std::vector<Case> r; return r; }
struct Registrar { Registrar(const char *s, const char *n, void (*f)()) { registry().push_back({s, n, f}); } };
// `ASSERT_X(...) << "context"`: the message is collected, the throw happens when the temporary dies
struct Thrower {
    std::ostringstream msg;
    Thrower(const char *file, int line, const std::string &what) { msg << file << ":" << line << ": " << what; }
    Thrower(Thrower &&o) : msg(std::move(o.msg)) {}
    template <class T> Thrower &operator<<(const T &v) { msg << v; return *this; }
    ~Thrower() noexcept(false) { throw Failure(msg.str()); }
};
struct Voidify { void operator&(const Thrower &) const {} };
// values without an operator<< print as a byte count (googletest prints their bytes)
template <class T, class = void> struct Streamable : std::false_type {};
template <class T> struct Streamable<T, std::void_t<decltype(std::declval<std::ostream &>() << std::declval<const T &>())>> : std::true_type {};
template <class T> inline std::string show(const T &v) {
    if constexpr (std::is_same<T, std::nullptr_t>::value) return "nullptr";
    else if constexpr (Streamable<T>::value) { std::ostringstream s; s << v; return s.str(); }
    else return "<" + std::to_string(sizeof(T)) + "-byte object>";
}
// ASSERT_DOUBLE_EQ: within 4 units in the last place (googletest's definition)
inline bool almost_equal(double a, double b) {
    if (std::isnan(a) || std::isnan(b)) return false;
    if (a == b) return true;
    int64_t ia, ib;
    std::memcpy(&ia, &a, 8);
    std::memcpy(&ib, &b, 8);
    auto biased = [](int64_t s) { return s < 0 ? (uint64_t)(~s + 1) : (uint64_t)s | 0x8000000000000000ull; };
    const uint64_t ua = biased(ia), ub = biased(ib);
    return (ua > ub ? ua - ub : ub - ua) <= 4;
}
inline int run_all(int argc, char **argv) {
    std::string filter;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.rfind("--gtest_filter=", 0) == 0) a = a.substr(15);
        if (!a.empty() && a[0] != '-') filter = a;
    }
    int ran = 0, failed = 0;
    for (const Case &c : registry()) {
        const std::string full = std::string(c.suite) + "." + c.name;
        if (!filter.empty() && full.find(filter) == std::string::npos) continue;
        std::cout << "[ RUN      ] " << full << std::endl;
        const auto t0 = std::chrono::steady_clock::now();
        bool ok = true;
        try { c.fn(); }
        catch (const Failure &f) { ok = false; std::cout << f.what() << std::endl; }
        catch (const std::exception &e) { ok = false; std::cout << "unexpected exception: " << e.what() << std::endl; }
        const long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        std::cout << (ok ? "[       OK ] " : "[  FAILED  ] ") << full << " (" << ms << " ms)" << std::endl;
        ran++;
        failed += !ok;
    }
    std::cout << "[==========] " << ran << " tests ran, " << failed << " failed." << std::endl;
    if (ran && !failed) std::cout << "[  PASSED  ] " << ran << " tests." << std::endl;
    return failed || !ran ? 1 : 0;
}
} // namespace gtest_shim

#define TEST(suite, name)                                                                                  \
    static void suite##_##name##_body();                                                                   \
    static ::gtest_shim::Registrar suite##_##name##_reg(#suite, #name, &suite##_##name##_body);            \
    static void suite##_##name##_body()

#define GTEST_SHIM_FAIL_(what) ::gtest_shim::Voidify() & ::gtest_shim::Thrower(__FILE__, __LINE__, what)
#define GTEST_SHIM_CHECK_(cond, what) if (cond) ; else GTEST_SHIM_FAIL_(what)
#define GTEST_SHIM_CMP_(a, op, b)                                                                          \
    if (auto &&gs_a_ = (a); true) if (auto &&gs_b_ = (b); gs_a_ op gs_b_) ; else                           \
        GTEST_SHIM_FAIL_(std::string("expected ") + #a + " " #op " " #b + ", got " + ::gtest_shim::show(gs_a_) + " vs " + ::gtest_shim::show(gs_b_))

#define ASSERT_TRUE(c) GTEST_SHIM_CHECK_(static_cast<bool>(c), std::string("expected true: ") + #c)
#define ASSERT_FALSE(c) GTEST_SHIM_CHECK_(!static_cast<bool>(c), std::string("expected false: ") + #c)
#define ASSERT_EQ(a, b) GTEST_SHIM_CMP_(a, ==, b)
#define ASSERT_NE(a, b) GTEST_SHIM_CMP_(a, !=, b)
#define ASSERT_LT(a, b) GTEST_SHIM_CMP_(a, <, b)
#define ASSERT_LE(a, b) GTEST_SHIM_CMP_(a, <=, b)
#define ASSERT_GT(a, b) GTEST_SHIM_CMP_(a, >, b)
#define ASSERT_GE(a, b) GTEST_SHIM_CMP_(a, >=, b)
#define ASSERT_STREQ(a, b) GTEST_SHIM_CHECK_(std::string(a) == std::string(b), std::string("expected equal strings: ") + #a + ", " #b)
#define ASSERT_NEAR(a, b, tol) GTEST_SHIM_CHECK_(std::fabs((double)(a) - (double)(b)) <= (double)(tol), std::string("expected |") + #a + " - " #b "| <= " #tol + ", got " + ::gtest_shim::show((double)(a)) + " vs " + ::gtest_shim::show((double)(b)))
#define ASSERT_DOUBLE_EQ(a, b) GTEST_SHIM_CHECK_(::gtest_shim::almost_equal((double)(a), (double)(b)), std::string("expected ") + #a + " ~ " #b + ", got " + ::gtest_shim::show((double)(a)) + " vs " + ::gtest_shim::show((double)(b)))
#define ASSERT_NO_THROW(stmt) do { try { stmt; } catch (...) { GTEST_SHIM_FAIL_(std::string("expected no throw: ") + #stmt); } } while (0)
#define ASSERT_THROW(stmt, exc) do { bool gs_t_ = false; try { stmt; } catch (const exc &) { gs_t_ = true; } catch (...) {} GTEST_SHIM_CHECK_(gs_t_, std::string("expected ") + #exc + " from: " #stmt); } while (0)
#define ASSERT_ANY_THROW(stmt) do { bool gs_t_ = false; try { stmt; } catch (...) { gs_t_ = true; } GTEST_SHIM_CHECK_(gs_t_, std::string("expected a throw from: ") + #stmt); } while (0)
#define EXPECT_TRUE ASSERT_TRUE
#define EXPECT_FALSE ASSERT_FALSE
#define EXPECT_EQ ASSERT_EQ
#define EXPECT_NE ASSERT_NE
#define EXPECT_NEAR ASSERT_NEAR
#define EXPECT_DOUBLE_EQ ASSERT_DOUBLE_EQ
