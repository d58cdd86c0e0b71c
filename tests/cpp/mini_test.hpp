// Tiny test harness for the C++ mirrors of the reference's in-file #[test] functions.
#pragma once
#include <cmath>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

struct TestCase { const char* name; std::function<void()> fn; };
inline std::vector<TestCase>& registry() { static std::vector<TestCase> r; return r; }
struct Registrar { Registrar(const char* n, std::function<void()> f) { registry().push_back({n, std::move(f)}); } };
#define TEST(name) static void name(); static Registrar reg_##name(#name, name); static void name()
struct Failure { std::string msg; };
#define CHECK(cond) do { if (!(cond)) throw Failure{std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": CHECK(" #cond ") failed"}; } while (0)
#define CHECK_EQ(a, b) do { if (!((a) == (b))) throw Failure{std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " #a " != " #b}; } while (0)
#define CHECK_NEAR(a, b, rel) do { const double _x = (a), _y = (b); if (!(std::fabs(_x - _y) <= (rel) * std::fabs(_y) + 1e-300)) \
    throw Failure{std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " #a " = " + std::to_string(_x) + " vs " + std::to_string(_y)}; } while (0)
#define CHECK_THROWS(expr) do { bool _t = false; try { (void)(expr); } catch (const std::exception&) { _t = true; } \
    if (!_t) throw Failure{std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": expected an error from " #expr}; } while (0)

inline int run_all() {
    int failed = 0;
    for (auto& t : registry()) {
        try { t.fn(); std::printf("ok   %s\n", t.name); }
        catch (const Failure& f) { ++failed; std::printf("FAIL %s: %s\n", t.name, f.msg.c_str()); }
        catch (const std::exception& e) { ++failed; std::printf("FAIL %s: exception: %s\n", t.name, e.what()); }
    }
    std::printf("%zu tests, %d failed\n", registry().size(), failed);
    return failed ? 1 : 0;
}
