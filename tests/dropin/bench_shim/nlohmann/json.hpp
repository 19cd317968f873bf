// TEST / BENCHMARK INFRASTRUCTURE.  Stand-in for <nlohmann/json.hpp> (fetched by the
// reference's CMake at configure time, not in this image) so that Ginkgo's own benchmark
// drivers (benchmark/spmv/spmv.cpp, benchmark/solver/solver.cpp) compile UNMODIFIED against the
// drop-in backend (oracle/build_benchmarks.py).  Implements the subset of nlohmann::ordered_json
// those sources use: null / bool / integer / float / string / array / insertion-ordered object,
// parse, dump / operator<< (std::setw = indentation), operator[] / at / contains / size / empty,
// is_*, get<T>, push_back / emplace_back, iteration over arrays and objects (items()).
// Written from the library's public documentation; no nlohmann code.
#ifndef GKO_CDNA4_JSON_SHIM_HPP_
#define GKO_CDNA4_JSON_SHIM_HPP_

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <iomanip>
#include <istream>
#include <iterator>
#include <limits>
#include <map>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace nlohmann {

class ordered_json {
public:
    enum class value_t { null, boolean, number_integer, number_unsigned, number_float, string, array, object };
    using array_t = std::vector<ordered_json>;
    using object_t = std::vector<std::pair<std::string, ordered_json>>;
    struct parse_error : std::runtime_error {
        using std::runtime_error::runtime_error;
    };
    struct type_error : std::runtime_error {
        using std::runtime_error::runtime_error;
    };
    struct out_of_range : std::out_of_range {
        using std::out_of_range::out_of_range;
    };
    using exception = std::exception;

    // ---- construction
    ordered_json() = default;
    ordered_json(std::nullptr_t) {}
    ordered_json(bool b) : type_(value_t::boolean), bool_(b) {}
    template <typename T, std::enable_if_t<std::is_integral<T>::value && std::is_signed<T>::value &&
                                               !std::is_same<T, bool>::value,
                                           int> = 0>
    ordered_json(T v) : type_(value_t::number_integer), int_(static_cast<std::int64_t>(v))
    {}
    template <typename T, std::enable_if_t<std::is_integral<T>::value && std::is_unsigned<T>::value &&
                                               !std::is_same<T, bool>::value,
                                           long> = 0>
    ordered_json(T v) : type_(value_t::number_unsigned), uint_(static_cast<std::uint64_t>(v))
    {}
    template <typename T, std::enable_if_t<std::is_floating_point<T>::value, char> = 0>
    ordered_json(T v) : type_(value_t::number_float), float_(static_cast<double>(v))
    {}
    ordered_json(const char* s) : type_(value_t::string), str_(s) {}
    ordered_json(const std::string& s) : type_(value_t::string), str_(s) {}
    ordered_json(std::string&& s) : type_(value_t::string), str_(std::move(s)) {}
    template <typename T>
    ordered_json(const std::vector<T>& v) : type_(value_t::array)
    {
        for (const auto& e : v) arr_.emplace_back(e);
    }

    static ordered_json object()
    {
        ordered_json j;
        j.type_ = value_t::object;
        return j;
    }
    static ordered_json array()
    {
        ordered_json j;
        j.type_ = value_t::array;
        return j;
    }

    // ---- type queries
    value_t type() const { return type_; }
    bool is_null() const { return type_ == value_t::null; }
    bool is_boolean() const { return type_ == value_t::boolean; }
    bool is_number_integer() const
    {
        return type_ == value_t::number_integer || type_ == value_t::number_unsigned;
    }
    bool is_number_unsigned() const { return type_ == value_t::number_unsigned; }
    bool is_number_float() const { return type_ == value_t::number_float; }
    bool is_number() const { return is_number_integer() || is_number_float(); }
    bool is_string() const { return type_ == value_t::string; }
    bool is_array() const { return type_ == value_t::array; }
    bool is_object() const { return type_ == value_t::object; }
    bool is_primitive() const { return !is_array() && !is_object(); }

    // ---- element access
    ordered_json& operator[](const std::string& key)
    {
        if (is_null()) type_ = value_t::object;
        need(value_t::object, "operator[] with a string key");
        for (auto& kv : obj_) {
            if (kv.first == key) return kv.second;
        }
        obj_.emplace_back(key, ordered_json{});
        return obj_.back().second;
    }
    ordered_json& operator[](const char* key) { return (*this)[std::string(key)]; }
    const ordered_json& operator[](const std::string& key) const { return at(key); }
    const ordered_json& operator[](const char* key) const { return at(std::string(key)); }
    template <typename I, std::enable_if_t<std::is_integral<I>::value, int> = 0>
    ordered_json& operator[](I idx)
    {
        if (is_null()) type_ = value_t::array;
        need(value_t::array, "operator[] with an index");
        if (static_cast<std::size_t>(idx) >= arr_.size()) arr_.resize(static_cast<std::size_t>(idx) + 1);
        return arr_[static_cast<std::size_t>(idx)];
    }
    template <typename I, std::enable_if_t<std::is_integral<I>::value, int> = 0>
    const ordered_json& operator[](I idx) const
    {
        return at(static_cast<std::size_t>(idx));
    }
    ordered_json& at(const std::string& key)
    {
        need(value_t::object, "at(key)");
        for (auto& kv : obj_) {
            if (kv.first == key) return kv.second;
        }
        throw out_of_range("json: key '" + key + "' not found");
    }
    const ordered_json& at(const std::string& key) const
    {
        return const_cast<ordered_json*>(this)->at(key);
    }
    ordered_json& at(std::size_t idx)
    {
        need(value_t::array, "at(index)");
        if (idx >= arr_.size()) throw out_of_range("json: array index out of range");
        return arr_[idx];
    }
    const ordered_json& at(std::size_t idx) const { return const_cast<ordered_json*>(this)->at(idx); }
    bool contains(const std::string& key) const
    {
        if (!is_object()) return false;
        for (const auto& kv : obj_) {
            if (kv.first == key) return true;
        }
        return false;
    }
    std::size_t count(const std::string& key) const { return contains(key) ? 1 : 0; }
    std::size_t erase(const std::string& key)
    {
        need(value_t::object, "erase(key)");
        for (auto it = obj_.begin(); it != obj_.end(); ++it) {
            if (it->first == key) {
                obj_.erase(it);
                return 1;
            }
        }
        return 0;
    }
    std::size_t size() const
    {
        return is_array() ? arr_.size() : is_object() ? obj_.size() : is_null() ? 0 : 1;
    }
    bool empty() const { return size() == 0; }
    void clear()
    {
        arr_.clear();
        obj_.clear();
        str_.clear();
    }

    // ---- modifiers
    void push_back(const ordered_json& v)
    {
        if (is_null()) type_ = value_t::array;
        need(value_t::array, "push_back");
        arr_.push_back(v);
    }
    void push_back(ordered_json&& v)
    {
        if (is_null()) type_ = value_t::array;
        need(value_t::array, "push_back");
        arr_.push_back(std::move(v));
    }
    template <typename... Args>
    ordered_json& emplace_back(Args&&... args)
    {
        if (is_null()) type_ = value_t::array;
        need(value_t::array, "emplace_back");
        arr_.emplace_back(std::forward<Args>(args)...);
        return arr_.back();
    }

    // ---- conversions
    template <typename T>
    T get() const
    {
        return get_impl(static_cast<T*>(nullptr));
    }
    template <typename T>
    void get_to(T& out) const
    {
        out = get<T>();
    }
    template <typename T, std::enable_if_t<!std::is_same<T, ordered_json>::value &&
                                               !std::is_same<T, std::initializer_list<ordered_json>>::value &&
                                               !std::is_pointer<T>::value &&
                                               !std::is_same<T, char>::value,
                                           int> = 0>
    operator T() const
    {
        return get<T>();
    }

    // ---- iteration: arrays and objects yield their values; items() yields key + value
    class iterator {
    public:
        using iterator_category = std::forward_iterator_tag;
        using value_type = ordered_json;
        using difference_type = std::ptrdiff_t;
        using pointer = ordered_json*;
        using reference = ordered_json&;
        iterator(ordered_json* owner, std::size_t pos) : owner_(owner), pos_(pos) {}
        reference operator*() const { return value(); }
        pointer operator->() const { return &value(); }
        iterator& operator++()
        {
            ++pos_;
            return *this;
        }
        iterator operator++(int)
        {
            iterator t = *this;
            ++pos_;
            return t;
        }
        bool operator==(const iterator& o) const { return owner_ == o.owner_ && pos_ == o.pos_; }
        bool operator!=(const iterator& o) const { return !(*this == o); }
        const std::string& key() const { return owner_->obj_[pos_].first; }
        reference value() const
        {
            return owner_->is_object() ? owner_->obj_[pos_].second : owner_->arr_[pos_];
        }

    private:
        ordered_json* owner_;
        std::size_t pos_;
    };
    using const_iterator = iterator;
    iterator begin() { return iterator(this, 0); }
    iterator end() { return iterator(this, is_object() ? obj_.size() : arr_.size()); }
    iterator begin() const { return iterator(const_cast<ordered_json*>(this), 0); }
    iterator end() const
    {
        return iterator(const_cast<ordered_json*>(this), is_object() ? obj_.size() : arr_.size());
    }
    iterator find(const std::string& key)
    {
        if (is_object()) {
            for (std::size_t i = 0; i < obj_.size(); ++i) {
                if (obj_[i].first == key) return iterator(this, i);
            }
        }
        return end();
    }
    iterator find(const std::string& key) const { return const_cast<ordered_json*>(this)->find(key); }
    struct items_proxy {
        ordered_json* j;
        struct it {
            iterator base;
            it& operator++()
            {
                ++base;
                return *this;
            }
            bool operator!=(const it& o) const { return base != o.base; }
            const it& operator*() const { return *this; }
            const std::string& key() const { return base.key(); }
            ordered_json& value() const { return base.value(); }
        };
        it begin() const { return it{j->begin()}; }
        it end() const { return it{j->end()}; }
    };
    items_proxy items() { return items_proxy{this}; }
    items_proxy items() const { return items_proxy{const_cast<ordered_json*>(this)}; }

    // ---- comparison
    friend bool operator==(const ordered_json& a, const ordered_json& b)
    {
        if (a.is_number() && b.is_number()) {
            if (a.is_number_float() || b.is_number_float()) return a.as_double() == b.as_double();
            return a.as_int() == b.as_int();
        }
        if (a.type_ != b.type_) return false;
        switch (a.type_) {
        case value_t::null: return true;
        case value_t::boolean: return a.bool_ == b.bool_;
        case value_t::string: return a.str_ == b.str_;
        case value_t::array: return a.arr_ == b.arr_;
        case value_t::object: return a.obj_ == b.obj_;
        default: return false;
        }
    }
    friend bool operator!=(const ordered_json& a, const ordered_json& b) { return !(a == b); }
    template <typename T, std::enable_if_t<!std::is_same<T, ordered_json>::value, int> = 0>
    friend bool operator==(const ordered_json& a, const T& b)
    {
        return a == ordered_json(b);
    }
    template <typename T, std::enable_if_t<!std::is_same<T, ordered_json>::value, int> = 0>
    friend bool operator!=(const ordered_json& a, const T& b)
    {
        return !(a == ordered_json(b));
    }

    // ordering against numbers and between numbers / strings
#define GKO_SHIM_JSON_ORDER_(op)                                                                 \
    template <typename T, std::enable_if_t<std::is_arithmetic<T>::value, int> = 0>               \
    friend bool operator op(const ordered_json& a, T b)                                          \
    {                                                                                            \
        return a.as_double() op static_cast<double>(b);                                          \
    }                                                                                            \
    template <typename T, std::enable_if_t<std::is_arithmetic<T>::value, int> = 0>               \
    friend bool operator op(T a, const ordered_json& b)                                          \
    {                                                                                            \
        return static_cast<double>(a) op b.as_double();                                          \
    }                                                                                            \
    friend bool operator op(const ordered_json& a, const ordered_json& b)                        \
    {                                                                                            \
        if (a.is_string() && b.is_string()) return a.str_ op b.str_;                             \
        return a.as_double() op b.as_double();                                                   \
    }
    GKO_SHIM_JSON_ORDER_(<)
    GKO_SHIM_JSON_ORDER_(>)
    GKO_SHIM_JSON_ORDER_(<=)
    GKO_SHIM_JSON_ORDER_(>=)
#undef GKO_SHIM_JSON_ORDER_

    // ---- text
    std::string dump(int indent = -1) const
    {
        std::string out;
        write(out, indent, 0);
        return out;
    }
    friend std::ostream& operator<<(std::ostream& os, const ordered_json& j)
    {
        const int indent = os.width() > 0 ? static_cast<int>(os.width()) : -1;
        os.width(0);
        return os << j.dump(indent);
    }
    static ordered_json parse(const std::string& text)
    {
        parser p{text, 0};
        ordered_json j = p.value();
        p.skip();
        if (p.pos != text.size()) throw parse_error("json: trailing characters at " + std::to_string(p.pos));
        return j;
    }
    static ordered_json parse(std::istream& is)
    {
        std::stringstream ss;
        ss << is.rdbuf();
        return parse(ss.str());
    }
    static ordered_json parse(std::istream&& is) { return parse(is); }
    static ordered_json parse(const char* text) { return parse(std::string(text)); }
    friend std::istream& operator>>(std::istream& is, ordered_json& j)
    {
        j = parse(is);
        return is;
    }

private:
    value_t type_ = value_t::null;
    bool bool_ = false;
    std::int64_t int_ = 0;
    std::uint64_t uint_ = 0;
    double float_ = 0.0;
    std::string str_;
    array_t arr_;
    object_t obj_;

    void need(value_t t, const char* what) const
    {
        if (type_ != t) throw type_error(std::string("json: ") + what + " on a value of another type");
    }
    double as_double() const
    {
        switch (type_) {
        case value_t::number_float: return float_;
        case value_t::number_integer: return static_cast<double>(int_);
        case value_t::number_unsigned: return static_cast<double>(uint_);
        case value_t::boolean: return bool_ ? 1.0 : 0.0;
        default: throw type_error("json: value is not a number");
        }
    }
    std::int64_t as_int() const
    {
        switch (type_) {
        case value_t::number_float: return static_cast<std::int64_t>(float_);
        case value_t::number_integer: return int_;
        case value_t::number_unsigned: return static_cast<std::int64_t>(uint_);
        case value_t::boolean: return bool_ ? 1 : 0;
        default: throw type_error("json: value is not a number");
        }
    }
    // get<T> by overload on a null pointer of the target type
    ordered_json get_impl(ordered_json*) const { return *this; }
    bool get_impl(bool*) const
    {
        if (type_ == value_t::boolean) return bool_;
        throw type_error("json: value is not a boolean");
    }
    std::string get_impl(std::string*) const
    {
        need(value_t::string, "get<std::string>");
        return str_;
    }
    template <typename T, std::enable_if_t<std::is_integral<T>::value && !std::is_same<T, bool>::value, int> = 0>
    T get_impl(T*) const
    {
        if (type_ == value_t::number_unsigned) return static_cast<T>(uint_);
        return static_cast<T>(as_int());
    }
    template <typename T, std::enable_if_t<std::is_floating_point<T>::value, int> = 0>
    T get_impl(T*) const
    {
        return static_cast<T>(as_double());
    }
    template <typename T>
    std::vector<T> get_impl(std::vector<T>*) const
    {
        need(value_t::array, "get<std::vector>");
        std::vector<T> v;
        for (const auto& e : arr_) v.push_back(e.template get<T>());
        return v;
    }
    template <typename T>
    std::map<std::string, T> get_impl(std::map<std::string, T>*) const
    {
        need(value_t::object, "get<std::map>");
        std::map<std::string, T> m;
        for (const auto& kv : obj_) m[kv.first] = kv.second.template get<T>();
        return m;
    }

    static void write_string(std::string& out, const std::string& s)
    {
        out += '"';
        for (unsigned char c : s) {
            switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            default:
                if (c < 0x20) {
                    char buf[8];
                    std::snprintf(buf, sizeof(buf), "\\u%04x", c);
                    out += buf;
                } else {
                    out += static_cast<char>(c);
                }
            }
        }
        out += '"';
    }
    static void write_double(std::string& out, double v)
    {
        if (!std::isfinite(v)) {
            out += "null";
            return;
        }
        char buf[40];
        // shortest representation that round-trips
        for (int prec = 15; prec <= 17; ++prec) {
            std::snprintf(buf, sizeof(buf), "%.*g", prec, v);
            if (std::strtod(buf, nullptr) == v) break;
        }
        std::string s = buf;
        if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
        out += s;
    }
    void write(std::string& out, int indent, int depth) const
    {
        const bool pretty = indent >= 0;
        auto newline = [&](int d) {
            if (pretty) {
                out += '\n';
                out.append(static_cast<std::size_t>(indent) * d, ' ');
            }
        };
        switch (type_) {
        case value_t::null: out += "null"; break;
        case value_t::boolean: out += bool_ ? "true" : "false"; break;
        case value_t::number_integer: out += std::to_string(int_); break;
        case value_t::number_unsigned: out += std::to_string(uint_); break;
        case value_t::number_float: write_double(out, float_); break;
        case value_t::string: write_string(out, str_); break;
        case value_t::array:
            if (arr_.empty()) {
                out += "[]";
                break;
            }
            out += '[';
            for (std::size_t i = 0; i < arr_.size(); ++i) {
                if (i) out += ',';
                newline(depth + 1);
                arr_[i].write(out, indent, depth + 1);
            }
            newline(depth);
            out += ']';
            break;
        case value_t::object:
            if (obj_.empty()) {
                out += "{}";
                break;
            }
            out += '{';
            for (std::size_t i = 0; i < obj_.size(); ++i) {
                if (i) out += ',';
                newline(depth + 1);
                write_string(out, obj_[i].first);
                out += pretty ? ": " : ":";
                obj_[i].second.write(out, indent, depth + 1);
            }
            newline(depth);
            out += '}';
            break;
        }
    }

    struct parser {
        const std::string& s;
        std::size_t pos;
        void skip()
        {
            while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\n' || s[pos] == '\t' || s[pos] == '\r')) ++pos;
        }
        [[noreturn]] void fail(const char* what) const
        {
            throw parse_error(std::string("json parse error at ") + std::to_string(pos) + ": " + what);
        }
        bool take(const char* lit)
        {
            const std::size_t n = std::char_traits<char>::length(lit);
            if (s.compare(pos, n, lit) == 0) {
                pos += n;
                return true;
            }
            return false;
        }
        std::string string()
        {
            if (s[pos] != '"') fail("expected a string");
            ++pos;
            std::string out;
            while (pos < s.size() && s[pos] != '"') {
                char c = s[pos++];
                if (c != '\\') {
                    out += c;
                    continue;
                }
                if (pos >= s.size()) fail("bad escape");
                c = s[pos++];
                switch (c) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    if (pos + 4 > s.size()) fail("bad \\u escape");
                    const unsigned cp = static_cast<unsigned>(std::strtoul(s.substr(pos, 4).c_str(), nullptr, 16));
                    pos += 4;
                    if (cp < 0x80) {
                        out += static_cast<char>(cp);
                    } else if (cp < 0x800) {
                        out += static_cast<char>(0xC0 | (cp >> 6));
                        out += static_cast<char>(0x80 | (cp & 0x3F));
                    } else {
                        out += static_cast<char>(0xE0 | (cp >> 12));
                        out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
                        out += static_cast<char>(0x80 | (cp & 0x3F));
                    }
                    break;
                }
                default: out += c;
                }
            }
            if (pos >= s.size()) fail("unterminated string");
            ++pos;
            return out;
        }
        ordered_json value()
        {
            skip();
            if (pos >= s.size()) fail("unexpected end of input");
            const char c = s[pos];
            if (c == '{') {
                ++pos;
                ordered_json j = ordered_json::object();
                skip();
                if (pos < s.size() && s[pos] == '}') {
                    ++pos;
                    return j;
                }
                for (;;) {
                    skip();
                    std::string key = string();
                    skip();
                    if (pos >= s.size() || s[pos] != ':') fail("expected ':'");
                    ++pos;
                    j[key] = value();
                    skip();
                    if (pos < s.size() && s[pos] == ',') {
                        ++pos;
                        continue;
                    }
                    if (pos < s.size() && s[pos] == '}') {
                        ++pos;
                        return j;
                    }
                    fail("expected ',' or '}'");
                }
            }
            if (c == '[') {
                ++pos;
                ordered_json j = ordered_json::array();
                skip();
                if (pos < s.size() && s[pos] == ']') {
                    ++pos;
                    return j;
                }
                for (;;) {
                    j.push_back(value());
                    skip();
                    if (pos < s.size() && s[pos] == ',') {
                        ++pos;
                        continue;
                    }
                    if (pos < s.size() && s[pos] == ']') {
                        ++pos;
                        return j;
                    }
                    fail("expected ',' or ']'");
                }
            }
            if (c == '"') return ordered_json(string());
            if (take("true")) return ordered_json(true);
            if (take("false")) return ordered_json(false);
            if (take("null")) return ordered_json();
            // number
            const std::size_t start = pos;
            if (s[pos] == '-') ++pos;
            bool is_float = false;
            while (pos < s.size() && (std::isdigit(static_cast<unsigned char>(s[pos])) || s[pos] == '.' ||
                                      s[pos] == 'e' || s[pos] == 'E' || s[pos] == '+' || s[pos] == '-')) {
                if (s[pos] == '.' || s[pos] == 'e' || s[pos] == 'E') is_float = true;
                ++pos;
            }
            if (pos == start) fail("unexpected character");
            const std::string num = s.substr(start, pos - start);
            if (is_float) return ordered_json(std::strtod(num.c_str(), nullptr));
            if (num[0] == '-') return ordered_json(static_cast<std::int64_t>(std::strtoll(num.c_str(), nullptr, 10)));
            return ordered_json(static_cast<std::uint64_t>(std::strtoull(num.c_str(), nullptr, 10)));
        }
    };
};

using json = ordered_json;

}  // namespace nlohmann

#endif  // GKO_CDNA4_JSON_SHIM_HPP_
