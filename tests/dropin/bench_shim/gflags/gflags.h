// TEST / BENCHMARK INFRASTRUCTURE.  Stand-in for <gflags/gflags.h> (fetched by the reference's
// CMake at configure time, not in this image): DEFINE_{bool,int32,uint32,int64,uint64,double,string},
// the FLAGS_* variables, gflags::ParseCommandLineFlags (-flag=value, --flag=value, -flag value,
// -boolflag, -noboolflag, --help), SetUsageMessage / SetVersionString.  Enough for Ginkgo's own
// benchmark drivers (benchmark/utils/general.hpp) to compile unmodified.  Own code.
#ifndef GKO_CDNA4_GFLAGS_SHIM_H_
#define GKO_CDNA4_GFLAGS_SHIM_H_

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <map>
#include <string>
#include <vector>

namespace gflags {

using int32 = std::int32_t;
using uint32 = std::uint32_t;
using int64 = std::int64_t;
using uint64 = std::uint64_t;

namespace detail {

struct flag_info {
    std::string type, help, default_text;
    std::function<bool(const std::string&)> set;
    bool is_bool = false;
};

inline std::map<std::string, flag_info>& registry()
{
    static std::map<std::string, flag_info> r;
    return r;
}
inline std::string& usage()
{
    static std::string u;
    return u;
}
inline std::string& version()
{
    static std::string v;
    return v;
}

inline bool parse_bool(const std::string& s, bool* out)
{
    if (s == "true" || s == "1" || s == "t" || s == "yes" || s == "y") {
        *out = true;
        return true;
    }
    if (s == "false" || s == "0" || s == "f" || s == "no" || s == "n") {
        *out = false;
        return true;
    }
    return false;
}

template <typename T>
struct registrar {
    registrar(const char* name, const char* type, T* var, const char* help, const std::string& def)
    {
        flag_info fi;
        fi.type = type;
        fi.help = help;
        fi.default_text = def;
        fi.is_bool = std::is_same<T, bool>::value;
        fi.set = [var](const std::string& text) { return assign(var, text); };
        registry()[name] = fi;
    }
    static bool assign(bool* v, const std::string& t) { return parse_bool(t, v); }
    static bool assign(std::string* v, const std::string& t)
    {
        *v = t;
        return true;
    }
    static bool assign(double* v, const std::string& t)
    {
        char* end = nullptr;
        *v = std::strtod(t.c_str(), &end);
        return end && *end == 0 && !t.empty();
    }
    template <typename I>
    static bool assign(I* v, const std::string& t)
    {
        char* end = nullptr;
        if (std::is_signed<I>::value) {
            *v = static_cast<I>(std::strtoll(t.c_str(), &end, 10));
        } else {
            *v = static_cast<I>(std::strtoull(t.c_str(), &end, 10));
        }
        return end && *end == 0 && !t.empty();
    }
};

}  // namespace detail

inline void SetUsageMessage(const std::string& s) { detail::usage() = s; }
inline void SetVersionString(const std::string& s) { detail::version() = s; }

inline void ShowUsageWithFlags(const char* argv0)
{
    std::cout << argv0 << ": " << detail::usage() << "\n\n  Flags:\n";
    for (const auto& kv : detail::registry()) {
        std::cout << "    -" << kv.first << " (" << kv.second.help << ") type: " << kv.second.type
                  << " default: " << kv.second.default_text << "\n";
    }
}

// removes the recognised flags from argv when remove_flags is set; returns the index of the
// first remaining argument
inline std::uint32_t ParseCommandLineFlags(int* argc, char*** argv, bool remove_flags)
{
    std::vector<char*> rest;
    rest.push_back((*argv)[0]);
    for (int i = 1; i < *argc; ++i) {
        std::string a = (*argv)[i];
        if (a == "--") {
            for (int k = i + 1; k < *argc; ++k) rest.push_back((*argv)[k]);
            break;
        }
        if (a.size() < 2 || a[0] != '-') {
            rest.push_back((*argv)[i]);
            continue;
        }
        std::string body = a.substr(a[1] == '-' ? 2 : 1);
        if (body == "help" || body == "helpfull" || body == "helpshort") {
            ShowUsageWithFlags((*argv)[0]);
            std::exit(0);
        }
        if (body == "version") {
            std::cout << detail::version() << std::endl;
            std::exit(0);
        }
        std::string name = body, value;
        bool has_value = false;
        const auto eq = body.find('=');
        if (eq != std::string::npos) {
            name = body.substr(0, eq);
            value = body.substr(eq + 1);
            has_value = true;
        }
        auto& reg = detail::registry();
        auto it = reg.find(name);
        if (it == reg.end() && !has_value && name.rfind("no", 0) == 0) {
            auto neg = reg.find(name.substr(2));
            if (neg != reg.end() && neg->second.is_bool) {
                neg->second.set("false");
                continue;
            }
        }
        if (it == reg.end()) {
            std::cerr << "ERROR: unknown command line flag '" << name << "'" << std::endl;
            std::exit(1);
        }
        if (!has_value) {
            if (it->second.is_bool) {
                value = "true";
            } else if (i + 1 < *argc) {
                value = (*argv)[++i];
            } else {
                std::cerr << "ERROR: flag '" << name << "' is missing its argument" << std::endl;
                std::exit(1);
            }
        }
        if (!it->second.set(value)) {
            std::cerr << "ERROR: illegal value '" << value << "' specified for " << it->second.type
                      << " flag '" << name << "'" << std::endl;
            std::exit(1);
        }
    }
    if (remove_flags) {
        for (std::size_t k = 0; k < rest.size(); ++k) (*argv)[k] = rest[k];
        *argc = static_cast<int>(rest.size());
        return 1;
    }
    return 1;
}

inline void ShutDownCommandLineFlags() {}

}  // namespace gflags

namespace google = gflags;

#define GKO_SHIM_DEFINE_FLAG_(ctype, tname, name, val, txt)                                   \
    ctype FLAGS_##name = val;                                                                 \
    static ::gflags::detail::registrar<ctype> gko_shim_flag_reg_##name(#name, tname, &FLAGS_##name, \
                                                                       txt, #val)
#define DEFINE_bool(name, val, txt) GKO_SHIM_DEFINE_FLAG_(bool, "bool", name, val, txt)
#define DEFINE_int32(name, val, txt) GKO_SHIM_DEFINE_FLAG_(::gflags::int32, "int32", name, val, txt)
#define DEFINE_uint32(name, val, txt) GKO_SHIM_DEFINE_FLAG_(::gflags::uint32, "uint32", name, val, txt)
#define DEFINE_int64(name, val, txt) GKO_SHIM_DEFINE_FLAG_(::gflags::int64, "int64", name, val, txt)
#define DEFINE_uint64(name, val, txt) GKO_SHIM_DEFINE_FLAG_(::gflags::uint64, "uint64", name, val, txt)
#define DEFINE_double(name, val, txt) GKO_SHIM_DEFINE_FLAG_(double, "double", name, val, txt)
#define DEFINE_string(name, val, txt) GKO_SHIM_DEFINE_FLAG_(std::string, "string", name, val, txt)
#define DECLARE_bool(name) extern bool FLAGS_##name
#define DECLARE_int32(name) extern ::gflags::int32 FLAGS_##name
#define DECLARE_uint32(name) extern ::gflags::uint32 FLAGS_##name
#define DECLARE_int64(name) extern ::gflags::int64 FLAGS_##name
#define DECLARE_uint64(name) extern ::gflags::uint64 FLAGS_##name
#define DECLARE_double(name) extern double FLAGS_##name
#define DECLARE_string(name) extern std::string FLAGS_##name

#endif  // GKO_CDNA4_GFLAGS_SHIM_H_
