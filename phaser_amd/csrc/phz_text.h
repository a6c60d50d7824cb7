// Small text helpers shared by the host-side writers / parsers: numbers printed the way Python prints them
// (str(int), repr(float) == str(numpy.float64)), field splitting, separator-joined string pools.
#ifndef PHZ_TEXT_H
#define PHZ_TEXT_H
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <charconv>
#include <string>
#include <string_view>
#include <vector>

namespace phztext {

inline void put_int(std::string &s, long long v) {
    char buf[24];
    auto r = std::to_chars(buf, buf + 24, v);
    s.append(buf, (size_t)(r.ptr - buf));
}

// repr(float) / str(numpy.float64): shortest round-trip digits, fixed notation for 1e-4 <= |x| < 1e16
inline void put_pyfloat(std::string &s, double x) {
    if (std::isnan(x)) { s += "nan"; return; }
    if (std::isinf(x)) { s += x < 0 ? "-inf" : "inf"; return; }
    if (x == 0) { s += std::signbit(x) ? "-0.0" : "0.0"; return; }
    char buf[64];
    auto r = std::to_chars(buf, buf + 64, x, std::chars_format::scientific);
    std::string_view v(buf, (size_t)(r.ptr - buf));
    if (v[0] == '-') { s += '-'; v.remove_prefix(1); }
    const size_t epos = v.find('e');
    std::string digits(1, v[0]);
    if (epos > 2) digits.append(v.substr(2, epos - 2));
    const int exp = atoi(std::string(v.substr(epos + 1)).c_str());
    if (exp >= -4 && exp < 16) {
        if (exp >= 0) {
            if ((int)digits.size() <= exp + 1) { s += digits; s.append((size_t)(exp + 1) - digits.size(), '0'); s += ".0"; }
            else { s.append(digits, 0, (size_t)exp + 1); s += '.'; s.append(digits, (size_t)exp + 1, std::string::npos); }
        } else {
            s += "0."; s.append((size_t)(-exp - 1), '0'); s += digits;
        }
    } else {
        s += digits[0];
        if (digits.size() > 1) { s += '.'; s.append(digits, 1, std::string::npos); }
        s += 'e'; s += exp < 0 ? '-' : '+';
        const int a = abs(exp);
        if (a < 10) s += '0';
        put_int(s, a);
    }
}

inline void split(std::string_view s, char sep, std::vector<std::string_view> &out) {
    out.clear();
    size_t i = 0;
    while (true) {
        size_t j = s.find(sep, i);
        if (j == std::string_view::npos) { out.push_back(s.substr(i)); break; }
        out.push_back(s.substr(i, j - i)); i = j + 1;
    }
}

// strings joined by one separator byte: item i = [off[i], off[i+1] - 1)
struct Pool {
    const uint32_t *off = nullptr;
    const char *b = nullptr;
    std::string_view at(int64_t i) const { return std::string_view(b + off[i], off[i + 1] - off[i] - 1); }
};

// offsets of a pool whose items are each followed by '\n'
inline std::vector<uint32_t> pool_offsets(const char *b, int64_t len) {
    std::vector<uint32_t> off(1, 0);
    for (int64_t i = 0; i < len; i++) if (b[i] == '\n') off.push_back((uint32_t)(i + 1));
    return off;
}

}  // namespace phztext
#endif
