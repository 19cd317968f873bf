// TEST INFRASTRUCTURE.  A small, self-contained stand-in for <gtest/gtest.h>:
// GoogleTest is fetched by the reference's CMake at configure time and is not in
// this image, so Ginkgo's own cross-executor test sources (test/matrix/*.cpp,
// test/solver/*.cpp, test/preconditioner/*.cpp, test/stop/*.cpp) are compiled
// UNMODIFIED against this header instead (oracle/build_reftests.py).  It implements
// the part of the GoogleTest interface those sources, core/test/utils.hpp,
// core/test/utils/assertions.hpp and core/test/gtest/*.cpp use: TEST / TEST_F /
// TYPED_TEST(_SUITE), ::testing::Test / Types / Environment / UnitTest / TestInfo,
// AssertionResult, ASSERT_* / EXPECT_* with message streaming, *_PRED_FORMATn,
// *_THROW, SCOPED_TRACE, GTEST_SKIP, PrintToString.  Written from the public
// GoogleTest documentation; no GoogleTest code.
#ifndef GKO_CDNA4_GTEST_SHIM_H_
#define GKO_CDNA4_GTEST_SHIM_H_

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <optional>
#include <ostream>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <type_traits>
#include <typeinfo>
#include <utility>
#include <vector>

namespace testing {

// ---- value printing -----------------------------------------------------------
namespace internal {

template <typename T, typename = void>
struct is_streamable : std::false_type {};
template <typename T>
struct is_streamable<T, decltype(void(std::declval<std::ostream&>() << std::declval<const T&>()))>
    : std::true_type {};

template <typename T, typename = void>
struct is_container : std::false_type {};
template <typename T>
struct is_container<T, decltype(void(std::declval<const T&>().begin()),
                                void(std::declval<const T&>().end()))> : std::true_type {};

template <typename T>
void print_value(std::ostream& os, const T& v);

template <typename T>
void print_dispatch(std::ostream& os, const T& v, std::true_type /*streamable*/, ...)
{
    os << v;
}
template <typename T>
void print_dispatch(std::ostream& os, const T& v, std::false_type, std::true_type /*container*/)
{
    os << "{ ";
    bool first = true;
    for (const auto& e : v) {
        if (!first) os << ", ";
        first = false;
        print_value(os, e);
    }
    os << " }";
}
template <typename T>
void print_dispatch(std::ostream& os, const T&, std::false_type, std::false_type)
{
    os << "<object of " << sizeof(T) << " bytes>";
}
inline void print_value(std::ostream& os, bool v) { os << (v ? "true" : "false"); }
inline void print_value(std::ostream& os, std::nullptr_t) { os << "nullptr"; }
inline void print_value(std::ostream& os, unsigned char v) { os << int(v); }
inline void print_value(std::ostream& os, signed char v) { os << int(v); }
inline void print_value(std::ostream& os, const std::string& v) { os << '"' << v << '"'; }
inline void print_value(std::ostream& os, const char* v) { os << (v ? v : "NULL"); }
template <typename A, typename B>
void print_value(std::ostream& os, const std::pair<A, B>& v)
{
    os << "(";
    print_value(os, v.first);
    os << ", ";
    print_value(os, v.second);
    os << ")";
}
template <typename T>
void print_value(std::ostream& os, const T& v)
{
    print_dispatch(os, v, is_streamable<T>{}, is_container<T>{});
}

}  // namespace internal

template <typename T>
std::string PrintToString(const T& v)
{
    std::ostringstream os;
    os << std::setprecision(std::numeric_limits<double>::digits10 + 2);
    internal::print_value(os, v);
    return os.str();
}

// ---- messages and assertion results ------------------------------------------
class Message {
public:
    Message() { ss_ << std::setprecision(std::numeric_limits<double>::digits10 + 2); }
    Message(const Message& o) : Message() { ss_ << o.str(); }
    template <typename T>
    Message& operator<<(const T& v)
    {
        ss_ << v;
        return *this;
    }
    Message& operator<<(std::ostream& (*manip)(std::ostream&))
    {
        ss_ << manip;
        return *this;
    }
    Message& operator<<(bool b)
    {
        ss_ << (b ? "true" : "false");
        return *this;
    }
    std::string str() const { return ss_.str(); }

private:
    std::ostringstream ss_;
};

class AssertionResult {
public:
    explicit AssertionResult(bool ok) : ok_(ok) {}
    AssertionResult(const AssertionResult&) = default;
    AssertionResult& operator=(const AssertionResult&) = default;
    explicit operator bool() const { return ok_; }
    AssertionResult operator!() const
    {
        AssertionResult r(!ok_);
        r.msg_ = msg_;
        return r;
    }
    const char* message() const { return msg_.c_str(); }
    const char* failure_message() const { return msg_.c_str(); }
    template <typename T>
    AssertionResult& operator<<(const T& v)
    {
        Message m;
        m << v;
        msg_ += m.str();
        return *this;
    }
    AssertionResult& operator<<(std::ostream& (*manip)(std::ostream&))
    {
        Message m;
        m << manip;
        msg_ += m.str();
        return *this;
    }
    friend bool operator==(const AssertionResult& a, const AssertionResult& b)
    {
        return a.ok_ == b.ok_;
    }
    friend bool operator!=(const AssertionResult& a, const AssertionResult& b)
    {
        return a.ok_ != b.ok_;
    }

private:
    bool ok_;
    std::string msg_;
};

inline AssertionResult AssertionSuccess() { return AssertionResult(true); }
inline AssertionResult AssertionFailure() { return AssertionResult(false); }
inline AssertionResult AssertionFailure(const Message& m) { return AssertionResult(false) << m.str(); }

// ---- tests, environments, registry --------------------------------------------
class Test {
public:
    virtual ~Test() = default;
    virtual void SetUp() {}
    virtual void TearDown() {}
    virtual void TestBody() = 0;
    static bool HasFatalFailure();
    static bool HasNonfatalFailure();
    static bool HasFailure();
    static bool IsSkipped();
    static void SetUpTestSuite() {}
    static void TearDownTestSuite() {}
};

class Environment {
public:
    virtual ~Environment() = default;
    virtual void SetUp() {}
    virtual void TearDown() {}
};

class TestInfo {
public:
    TestInfo(std::string suite, std::string name, std::string type_param)
        : suite_(std::move(suite)), name_(std::move(name)), type_param_(std::move(type_param))
    {}
    const char* test_suite_name() const { return suite_.c_str(); }
    const char* test_case_name() const { return suite_.c_str(); }
    const char* name() const { return name_.c_str(); }
    const char* type_param() const { return type_param_.empty() ? nullptr : type_param_.c_str(); }
    const char* value_param() const { return nullptr; }

private:
    std::string suite_, name_, type_param_;
};

namespace internal {

struct test_entry {
    std::string suite, name, type_param;
    std::function<Test*()> make;
};

struct state {
    std::vector<test_entry> tests;
    std::vector<Environment*> envs;
    std::vector<std::string> traces;
    const TestInfo* current = nullptr;
    bool fatal = false, nonfatal = false, skipped = false;
    std::string filter = "*";
    bool list_only = false;
    static state& get()
    {
        static state s;
        return s;
    }
};

inline bool register_test(std::string suite, std::string name, std::string type_param,
                          std::function<Test*()> make)
{
    state::get().tests.push_back({std::move(suite), std::move(name), std::move(type_param),
                                  std::move(make)});
    return true;
}

// glob with '*' and '?', alternatives separated by ':', negative part after '-'
inline bool glob_match(const char* p, const char* s)
{
    if (*p == 0) return *s == 0;
    if (*p == '*') return glob_match(p + 1, s) || (*s && glob_match(p, s + 1));
    return *s && (*p == '?' || *p == *s) && glob_match(p + 1, s + 1);
}
inline bool any_match(const std::string& patterns, const std::string& name)
{
    size_t b = 0;
    while (b <= patterns.size()) {
        size_t e = patterns.find(':', b);
        if (e == std::string::npos) e = patterns.size();
        if (e > b && glob_match(patterns.substr(b, e - b).c_str(), name.c_str())) return true;
        b = e + 1;
    }
    return false;
}
inline bool filter_match(const std::string& filter, const std::string& name)
{
    const size_t dash = filter.find('-');
    const std::string pos = dash == std::string::npos ? filter : filter.substr(0, dash);
    const std::string neg = dash == std::string::npos ? "" : filter.substr(dash + 1);
    return any_match(pos.empty() ? "*" : pos, name) && !(!neg.empty() && any_match(neg, name));
}

enum class kind { nonfatal, fatal, skip, success };

// receives the streamed user message and reports; operator= returns void so that
// "return AssertHelper(...) = Message() << ..." is valid in a void function
class AssertHelper {
public:
    AssertHelper(kind k, const char* file, int line, std::string text)
        : k_(k), file_(file), line_(line), text_(std::move(text))
    {}
    void operator=(const Message& m) const
    {
        state& s = state::get();
        if (k_ == kind::success) return;
        if (k_ == kind::skip) {
            s.skipped = true;
            std::cout << file_ << ":" << line_ << ": Skipped\n" << m.str() << std::endl;
            return;
        }
        (k_ == kind::fatal ? s.fatal : s.nonfatal) = true;
        std::cout << file_ << ":" << line_ << ": Failure\n" << text_;
        const std::string user = m.str();
        if (!user.empty()) std::cout << "\n" << user;
        for (auto it = s.traces.rbegin(); it != s.traces.rend(); ++it) {
            if (it == s.traces.rbegin()) std::cout << "\nGoogle Test trace:";
            std::cout << "\n" << *it;
        }
        std::cout << std::endl;
    }

private:
    kind k_;
    const char* file_;
    int line_;
    std::string text_;
};

class ScopedTrace {
public:
    template <typename T>
    ScopedTrace(const char* file, int line, const T& msg)
    {
        Message m;
        m << file << ":" << line << ": " << msg;
        state::get().traces.push_back(m.str());
    }
    ~ScopedTrace() { state::get().traces.pop_back(); }
};

template <typename T>
std::string type_name()
{
    return typeid(T).name();
}

// ---- comparison helpers ----
template <typename A, typename B>
AssertionResult cmp_failure(const char* ea, const char* eb, const A& a, const B& b, const char* op)
{
    return AssertionFailure() << "Expected: (" << ea << ") " << op << " (" << eb
                              << "), actual: " << PrintToString(a) << " vs " << PrintToString(b);
}
template <typename A, typename B>
AssertionResult cmp_eq(const char* ea, const char* eb, const A& a, const B& b)
{
    if (a == b) return AssertionSuccess();
    return AssertionFailure() << "Expected equality of these values:\n  " << ea << "\n    Which is: "
                              << PrintToString(a) << "\n  " << eb << "\n    Which is: "
                              << PrintToString(b);
}
#define GKO_SHIM_CMP_(name, op)                                                                \
    template <typename A, typename B>                                                          \
    AssertionResult cmp_##name(const char* ea, const char* eb, const A& a, const B& b)         \
    {                                                                                          \
        if (a op b) return AssertionSuccess();                                                 \
        return cmp_failure(ea, eb, a, b, #op);                                                 \
    }
GKO_SHIM_CMP_(ne, !=)
GKO_SHIM_CMP_(lt, <)
GKO_SHIM_CMP_(le, <=)
GKO_SHIM_CMP_(gt, >)
GKO_SHIM_CMP_(ge, >=)
#undef GKO_SHIM_CMP_

inline AssertionResult cmp_near(const char* ea, const char* eb, const char* et, double a, double b,
                                double tol)
{
    const double d = std::fabs(a - b);
    if (d <= tol) return AssertionSuccess();
    return AssertionFailure() << "The difference between " << ea << " and " << eb << " is " << d
                              << ", which exceeds " << et << ", where\n" << ea << " evaluates to "
                              << a << ",\n" << eb << " evaluates to " << b << ", and\n" << et
                              << " evaluates to " << tol << ".";
}
template <typename F>
AssertionResult cmp_almost(const char* ea, const char* eb, F a, F b)
{
    // 4 units in the last place, like GoogleTest
    if (a == b) return AssertionSuccess();
    const F scale = std::max(std::fabs(a), std::fabs(b));
    if (std::fabs(a - b) <= 4 * std::numeric_limits<F>::epsilon() * scale) return AssertionSuccess();
    return cmp_eq(ea, eb, a, b);
}
inline AssertionResult cmp_streq(const char* ea, const char* eb, const char* a, const char* b)
{
    if ((a == nullptr && b == nullptr) || (a && b && std::strcmp(a, b) == 0)) return AssertionSuccess();
    return cmp_eq(ea, eb, std::string(a ? a : "NULL"), std::string(b ? b : "NULL"));
}
inline AssertionResult bool_result(const AssertionResult& r, const char* expr, bool expected)
{
    if (bool(r) == expected) return AssertionSuccess();
    AssertionResult f = AssertionFailure();
    f << "Value of: " << expr << "\n  Actual: " << (expected ? "false" : "true");
    if (*r.message()) f << " (" << r.message() << ")";
    f << "\nExpected: " << (expected ? "true" : "false");
    return f;
}
inline AssertionResult bool_result(bool v, const char* expr, bool expected)
{
    return bool_result(AssertionResult(v), expr, expected);
}
template <typename T>
AssertionResult bool_result(const T& v, const char* expr, bool expected)
{
    return bool_result(AssertionResult(static_cast<bool>(v)), expr, expected);
}

// ---- typed tests ----
template <typename... Ts>
struct type_list {};

struct default_name_generator {
    template <typename T>
    static std::string GetName(int i)
    {
        return std::to_string(i);
    }
};

template <template <typename> class TestClass, typename NameGen, typename List>
struct typed_registrar;

template <template <typename> class TestClass, typename NameGen, template <typename...> class L,
          typename... Ts>
struct typed_registrar<TestClass, NameGen, L<Ts...>> {
    static bool go(const char* suite, const char* name)
    {
        int idx = 0;
        bool dummy[] = {true, reg<Ts>(suite, name, idx++)...};
        (void)dummy;
        return true;
    }
    template <typename T>
    static bool reg(const char* suite, const char* name, int idx)
    {
        return register_test(std::string(suite) + "/" + NameGen::template GetName<T>(idx), name,
                             type_name<T>(), [] { return static_cast<Test*>(new TestClass<T>); });
    }
};

template <typename... NameGen>
struct name_gen_select {
    using type = default_name_generator;
};
template <typename NameGen>
struct name_gen_select<NameGen> {
    using type = NameGen;
};

}  // namespace internal

template <typename... Ts>
struct Types {};

inline bool Test::HasFatalFailure() { return internal::state::get().fatal; }
inline bool Test::HasNonfatalFailure() { return internal::state::get().nonfatal; }
inline bool Test::HasFailure() { return HasFatalFailure() || HasNonfatalFailure(); }
inline bool Test::IsSkipped() { return internal::state::get().skipped; }

inline Environment* AddGlobalTestEnvironment(Environment* env)
{
    internal::state::get().envs.push_back(env);
    return env;
}

class UnitTest {
public:
    static UnitTest* GetInstance()
    {
        static UnitTest u;
        return &u;
    }
    const TestInfo* current_test_info() const { return internal::state::get().current; }
    int Run()
    {
        internal::state& s = internal::state::get();
        if (s.list_only) {
            for (const auto& t : s.tests) std::cout << t.suite << "." << t.name << "\n";
            return 0;
        }
        for (auto* e : s.envs) e->SetUp();
        int ran = 0, failed = 0, skipped = 0;
        std::vector<std::string> failed_names;
        for (const auto& t : s.tests) {
            const std::string full = t.suite + "." + t.name;
            if (!internal::filter_match(s.filter, full)) continue;
            TestInfo info(t.suite, t.name, t.type_param);
            s.current = &info;
            s.fatal = s.nonfatal = s.skipped = false;
            std::cout << "[ RUN      ] " << full << std::endl;
            try {
                std::unique_ptr<Test> test(t.make());
                if (!s.fatal && !s.skipped) test->SetUp();
                if (!s.fatal && !s.skipped) test->TestBody();
                test->TearDown();
            } catch (const std::exception& e) {
                s.fatal = true;
                std::cout << "unknown file: Failure\nC++ exception with description \"" << e.what()
                          << "\" thrown in the test body." << std::endl;
            } catch (...) {
                s.fatal = true;
                std::cout << "unknown file: Failure\nUnknown C++ exception thrown in the test body."
                          << std::endl;
            }
            ++ran;
            if (s.fatal || s.nonfatal) {
                ++failed;
                failed_names.push_back(full);
                std::cout << "[  FAILED  ] " << full << std::endl;
            } else if (s.skipped) {
                ++skipped;
                std::cout << "[  SKIPPED ] " << full << std::endl;
            } else {
                std::cout << "[       OK ] " << full << std::endl;
            }
            s.current = nullptr;
        }
        for (auto it = s.envs.rbegin(); it != s.envs.rend(); ++it) (*it)->TearDown();
        std::cout << "[==========] " << ran << " tests ran.\n[  PASSED  ] "
                  << ran - failed - skipped << " tests." << std::endl;
        if (skipped) std::cout << "[  SKIPPED ] " << skipped << " tests." << std::endl;
        if (failed) {
            std::cout << "[  FAILED  ] " << failed << " tests, listed below:" << std::endl;
            for (const auto& n : failed_names) std::cout << "[  FAILED  ] " << n << std::endl;
        }
        return failed ? 1 : 0;
    }
};

inline void InitGoogleTest(int* argc, char** argv)
{
    internal::state& s = internal::state::get();
    int out = 1;
    for (int i = 1; i < *argc; ++i) {
        const std::string a = argv[i];
        if (a.rfind("--gtest_filter=", 0) == 0) {
            s.filter = a.substr(15);
        } else if (a == "--gtest_list_tests") {
            s.list_only = true;
        } else if (a.rfind("--gtest_", 0) == 0) {
            // other GoogleTest flags: accepted, ignored
        } else {
            argv[out++] = argv[i];
        }
    }
    *argc = out;
}
inline void InitGoogleTest() {}

}  // namespace testing

// ---- macros --------------------------------------------------------------------
#define GKO_SHIM_CAT_(a, b) a##b
#define GKO_SHIM_CAT(a, b) GKO_SHIM_CAT_(a, b)
#define GKO_SHIM_CLASS_(suite, name) suite##_##name##_Test
#define GKO_SHIM_BLOCKER_ \
    switch (0)            \
    case 0:               \
    default:

#define RUN_ALL_TESTS() ::testing::UnitTest::GetInstance()->Run()

#define GKO_SHIM_TEST_(suite, name, parent)                                                      \
    class GKO_SHIM_CLASS_(suite, name) : public parent {                                         \
    public:                                                                                      \
        void TestBody() override;                                                                \
        static bool registered_;                                                                 \
    };                                                                                           \
    bool GKO_SHIM_CLASS_(suite, name)::registered_ = ::testing::internal::register_test(         \
        #suite, #name, "",                                                                       \
        [] { return static_cast<::testing::Test*>(new GKO_SHIM_CLASS_(suite, name)); });         \
    void GKO_SHIM_CLASS_(suite, name)::TestBody()

#define TEST(suite, name) GKO_SHIM_TEST_(suite, name, ::testing::Test)
#define TEST_F(fixture, name) GKO_SHIM_TEST_(fixture, name, fixture)

#define TYPED_TEST_SUITE(fixture, types, ...)                                  \
    typedef types gko_shim_types_##fixture##_;                                 \
    typedef ::testing::internal::name_gen_select<__VA_ARGS__>::type gko_shim_namegen_##fixture##_
#define TYPED_TEST_CASE TYPED_TEST_SUITE

#define TYPED_TEST(fixture, name)                                                                \
    template <typename gko_shim_TypeParam_>                                                      \
    class GKO_SHIM_CLASS_(fixture, name) : public fixture<gko_shim_TypeParam_> {                 \
    public:                                                                                      \
        typedef fixture<gko_shim_TypeParam_> TestFixture;                                        \
        typedef gko_shim_TypeParam_ TypeParam;                                                   \
        void TestBody() override;                                                                \
    };                                                                                           \
    static bool GKO_SHIM_CAT(gko_shim_reg_##fixture##_##name##_, __LINE__) =                     \
        ::testing::internal::typed_registrar<GKO_SHIM_CLASS_(fixture, name),                     \
                                             gko_shim_namegen_##fixture##_,                      \
                                             gko_shim_types_##fixture##_>::go(#fixture, #name);  \
    template <typename gko_shim_TypeParam_>                                                      \
    void GKO_SHIM_CLASS_(fixture, name)<gko_shim_TypeParam_>::TestBody()

// ---- assertion plumbing: `on_fail` is "return" (ASSERT_*) or empty (EXPECT_*) ----
#define GKO_SHIM_CHECK_(result_expr, fail_kind, on_fail)                                         \
    GKO_SHIM_BLOCKER_                                                                            \
    if (const ::testing::AssertionResult gko_shim_ar_ = (result_expr))                           \
        ;                                                                                        \
    else                                                                                         \
        on_fail ::testing::internal::AssertHelper(::testing::internal::kind::fail_kind, __FILE__, \
                                                  __LINE__, gko_shim_ar_.message()) =            \
            ::testing::Message()
#define GKO_SHIM_ASSERT_(r) GKO_SHIM_CHECK_(r, fatal, return)
#define GKO_SHIM_EXPECT_(r) GKO_SHIM_CHECK_(r, nonfatal, )

#define FAIL() GKO_SHIM_ASSERT_(::testing::AssertionFailure() << "Failed")
#define GTEST_FAIL() FAIL()
#define ADD_FAILURE() GKO_SHIM_EXPECT_(::testing::AssertionFailure() << "Failed")
#define SUCCEED() GKO_SHIM_EXPECT_(::testing::AssertionSuccess())
#define GTEST_SUCCEED() SUCCEED()
#define GTEST_SKIP()                                                                             \
    return ::testing::internal::AssertHelper(::testing::internal::kind::skip, __FILE__, __LINE__, \
                                             "") = ::testing::Message()

#define ASSERT_TRUE(c) GKO_SHIM_ASSERT_(::testing::internal::bool_result((c), #c, true))
#define ASSERT_FALSE(c) GKO_SHIM_ASSERT_(::testing::internal::bool_result((c), #c, false))
#define EXPECT_TRUE(c) GKO_SHIM_EXPECT_(::testing::internal::bool_result((c), #c, true))
#define EXPECT_FALSE(c) GKO_SHIM_EXPECT_(::testing::internal::bool_result((c), #c, false))

#define ASSERT_EQ(a, b) GKO_SHIM_ASSERT_(::testing::internal::cmp_eq(#a, #b, (a), (b)))
#define ASSERT_NE(a, b) GKO_SHIM_ASSERT_(::testing::internal::cmp_ne(#a, #b, (a), (b)))
#define ASSERT_LT(a, b) GKO_SHIM_ASSERT_(::testing::internal::cmp_lt(#a, #b, (a), (b)))
#define ASSERT_LE(a, b) GKO_SHIM_ASSERT_(::testing::internal::cmp_le(#a, #b, (a), (b)))
#define ASSERT_GT(a, b) GKO_SHIM_ASSERT_(::testing::internal::cmp_gt(#a, #b, (a), (b)))
#define ASSERT_GE(a, b) GKO_SHIM_ASSERT_(::testing::internal::cmp_ge(#a, #b, (a), (b)))
#define EXPECT_EQ(a, b) GKO_SHIM_EXPECT_(::testing::internal::cmp_eq(#a, #b, (a), (b)))
#define EXPECT_NE(a, b) GKO_SHIM_EXPECT_(::testing::internal::cmp_ne(#a, #b, (a), (b)))
#define EXPECT_LT(a, b) GKO_SHIM_EXPECT_(::testing::internal::cmp_lt(#a, #b, (a), (b)))
#define EXPECT_LE(a, b) GKO_SHIM_EXPECT_(::testing::internal::cmp_le(#a, #b, (a), (b)))
#define EXPECT_GT(a, b) GKO_SHIM_EXPECT_(::testing::internal::cmp_gt(#a, #b, (a), (b)))
#define EXPECT_GE(a, b) GKO_SHIM_EXPECT_(::testing::internal::cmp_ge(#a, #b, (a), (b)))

#define ASSERT_NEAR(a, b, t) GKO_SHIM_ASSERT_(::testing::internal::cmp_near(#a, #b, #t, (a), (b), (t)))
#define EXPECT_NEAR(a, b, t) GKO_SHIM_EXPECT_(::testing::internal::cmp_near(#a, #b, #t, (a), (b), (t)))
#define ASSERT_DOUBLE_EQ(a, b) GKO_SHIM_ASSERT_(::testing::internal::cmp_almost<double>(#a, #b, (a), (b)))
#define EXPECT_DOUBLE_EQ(a, b) GKO_SHIM_EXPECT_(::testing::internal::cmp_almost<double>(#a, #b, (a), (b)))
#define ASSERT_FLOAT_EQ(a, b) GKO_SHIM_ASSERT_(::testing::internal::cmp_almost<float>(#a, #b, (a), (b)))
#define EXPECT_FLOAT_EQ(a, b) GKO_SHIM_EXPECT_(::testing::internal::cmp_almost<float>(#a, #b, (a), (b)))
#define ASSERT_STREQ(a, b) GKO_SHIM_ASSERT_(::testing::internal::cmp_streq(#a, #b, (a), (b)))
#define EXPECT_STREQ(a, b) GKO_SHIM_EXPECT_(::testing::internal::cmp_streq(#a, #b, (a), (b)))

#define ASSERT_PRED_FORMAT1(p, a) GKO_SHIM_ASSERT_(p(#a, (a)))
#define ASSERT_PRED_FORMAT2(p, a, b) GKO_SHIM_ASSERT_(p(#a, #b, (a), (b)))
#define ASSERT_PRED_FORMAT3(p, a, b, c) GKO_SHIM_ASSERT_(p(#a, #b, #c, (a), (b), (c)))
#define ASSERT_PRED_FORMAT4(p, a, b, c, d) GKO_SHIM_ASSERT_(p(#a, #b, #c, #d, (a), (b), (c), (d)))
#define ASSERT_PRED_FORMAT5(p, a, b, c, d, e) \
    GKO_SHIM_ASSERT_(p(#a, #b, #c, #d, #e, (a), (b), (c), (d), (e)))
#define EXPECT_PRED_FORMAT1(p, a) GKO_SHIM_EXPECT_(p(#a, (a)))
#define EXPECT_PRED_FORMAT2(p, a, b) GKO_SHIM_EXPECT_(p(#a, #b, (a), (b)))
#define EXPECT_PRED_FORMAT3(p, a, b, c) GKO_SHIM_EXPECT_(p(#a, #b, #c, (a), (b), (c)))
#define EXPECT_PRED_FORMAT4(p, a, b, c, d) GKO_SHIM_EXPECT_(p(#a, #b, #c, #d, (a), (b), (c), (d)))
#define EXPECT_PRED_FORMAT5(p, a, b, c, d, e) \
    GKO_SHIM_EXPECT_(p(#a, #b, #c, #d, #e, (a), (b), (c), (d), (e)))

// statement-throws checks: run the statement inside a lambda, turn the outcome into
// an AssertionResult
#define GKO_SHIM_THROW_RESULT_(statement, exception_type)                                        \
    [&]() -> ::testing::AssertionResult {                                                        \
        try {                                                                                    \
            statement;                                                                           \
        } catch (const exception_type&) {                                                        \
            return ::testing::AssertionSuccess();                                                \
        } catch (const std::exception& gko_shim_e_) {                                            \
            return ::testing::AssertionFailure()                                                 \
                   << "Expected: " #statement " throws an exception of type " #exception_type    \
                      ".\n  Actual: it throws a different type (" << gko_shim_e_.what() << ").";  \
        } catch (...) {                                                                          \
            return ::testing::AssertionFailure()                                                 \
                   << "Expected: " #statement " throws an exception of type " #exception_type    \
                      ".\n  Actual: it throws a different type.";                                \
        }                                                                                        \
        return ::testing::AssertionFailure()                                                     \
               << "Expected: " #statement " throws an exception of type " #exception_type        \
                  ".\n  Actual: it throws nothing.";                                             \
    }()
#define GKO_SHIM_NO_THROW_RESULT_(statement)                                                     \
    [&]() -> ::testing::AssertionResult {                                                        \
        try {                                                                                    \
            statement;                                                                           \
        } catch (const std::exception& gko_shim_e_) {                                            \
            return ::testing::AssertionFailure()                                                 \
                   << "Expected: " #statement " doesn't throw an exception.\n  Actual: it "      \
                      "throws: " << gko_shim_e_.what();                                          \
        } catch (...) {                                                                          \
            return ::testing::AssertionFailure()                                                 \
                   << "Expected: " #statement " doesn't throw an exception.\n  Actual: it throws."; \
        }                                                                                        \
        return ::testing::AssertionSuccess();                                                    \
    }()
#define GKO_SHIM_ANY_THROW_RESULT_(statement)                                                    \
    [&]() -> ::testing::AssertionResult {                                                        \
        try {                                                                                    \
            statement;                                                                           \
        } catch (...) {                                                                          \
            return ::testing::AssertionSuccess();                                                \
        }                                                                                        \
        return ::testing::AssertionFailure()                                                     \
               << "Expected: " #statement " throws an exception.\n  Actual: it doesn't.";        \
    }()
#define ASSERT_THROW(s, t) GKO_SHIM_ASSERT_(GKO_SHIM_THROW_RESULT_(s, t))
#define EXPECT_THROW(s, t) GKO_SHIM_EXPECT_(GKO_SHIM_THROW_RESULT_(s, t))
#define ASSERT_NO_THROW(s) GKO_SHIM_ASSERT_(GKO_SHIM_NO_THROW_RESULT_(s))
#define EXPECT_NO_THROW(s) GKO_SHIM_EXPECT_(GKO_SHIM_NO_THROW_RESULT_(s))
#define ASSERT_ANY_THROW(s) GKO_SHIM_ASSERT_(GKO_SHIM_ANY_THROW_RESULT_(s))
#define EXPECT_ANY_THROW(s) GKO_SHIM_EXPECT_(GKO_SHIM_ANY_THROW_RESULT_(s))

#define SCOPED_TRACE(msg) \
    ::testing::internal::ScopedTrace GKO_SHIM_CAT(gko_shim_trace_, __LINE__)(__FILE__, __LINE__, (msg))

#endif  // GKO_CDNA4_GTEST_SHIM_H_
