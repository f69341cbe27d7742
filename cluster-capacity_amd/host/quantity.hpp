// quantity.hpp -- resource.Quantity -> integers, exactly.
//   Quantity.Value()       rounds up to an integer            (vendor/k8s.io/apimachinery/pkg/api/resource/quantity.go:813-820)
//   Quantity.MilliValue()  rounds up to an integer of 1/1000  (quantity.go:822-834)
// Grammar (quantity.go:52-80): <signedNumber><suffix>, suffix = binarySI (Ki Mi Gi Ti Pi Ei) | decimalSI (n u m "" k M G T P E)
// | decimalExponent (e<N> | E<N>).  Evaluated with 128-bit integers: mantissa * 10^e * 2^b, ceiling division for e < 0.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>

namespace cchost {

struct Quantity {
    __int128 mant = 0; // signed decimal digits without the point
    int exp10 = 0;     // value = mant * 10^exp10 * 2^exp2
    int exp2 = 0;
};

inline Quantity parse_quantity(const std::string &text) {
    size_t b = 0, e = text.size();
    while (b < e && (text[b] == ' ' || text[b] == '\t')) b++;
    while (e > b && (text[e - 1] == ' ' || text[e - 1] == '\t')) e--;
    const std::string s = text.substr(b, e - b);
    auto bad = [&]() -> std::runtime_error { return std::runtime_error("bad quantity '" + text + "'"); };
    size_t i = 0;
    bool neg = false;
    if (i < s.size() && (s[i] == '+' || s[i] == '-')) neg = s[i] == '-', i++;
    Quantity q;
    int digits = 0, frac = 0;
    bool dot = false;
    for (; i < s.size(); i++) {
        const char c = s[i];
        if (c >= '0' && c <= '9') {
            if (q.mant > ((__int128)1 << 100)) throw bad();
            q.mant = q.mant * 10 + (c - '0');
            digits++;
            if (dot) frac++;
        } else if (c == '.' && !dot)
            dot = true;
        else
            break;
    }
    if (digits == 0) throw bad();
    q.exp10 = -frac;
    const std::string suf = s.substr(i);
    if (!suf.empty() && (suf[0] == 'e' || suf[0] == 'E')) {
        size_t j = 1;
        bool eneg = false;
        if (j < suf.size() && (suf[j] == '+' || suf[j] == '-')) eneg = suf[j] == '-', j++;
        if (j >= suf.size()) {
            if (suf == "E") q.exp10 += 18; // the decimalSI suffix E (exa)
            else throw bad();
        } else {
            int ex = 0;
            for (; j < suf.size(); j++) {
                if (suf[j] < '0' || suf[j] > '9' || ex > 1000) throw bad();
                ex = ex * 10 + (suf[j] - '0');
            }
            q.exp10 += eneg ? -ex : ex;
        }
    } else if (suf == "Ki") q.exp2 = 10;
    else if (suf == "Mi") q.exp2 = 20;
    else if (suf == "Gi") q.exp2 = 30;
    else if (suf == "Ti") q.exp2 = 40;
    else if (suf == "Pi") q.exp2 = 50;
    else if (suf == "Ei") q.exp2 = 60;
    else if (suf == "n") q.exp10 -= 9;
    else if (suf == "u") q.exp10 -= 6;
    else if (suf == "m") q.exp10 -= 3;
    else if (suf.empty()) ;
    else if (suf == "k") q.exp10 += 3;
    else if (suf == "M") q.exp10 += 6;
    else if (suf == "G") q.exp10 += 9;
    else if (suf == "T") q.exp10 += 12;
    else if (suf == "P") q.exp10 += 15;
    else throw bad();
    if (neg) q.mant = -q.mant;
    return q;
}

// ceil(q * 10^scale10), saturating at the int64 range
inline int64_t quantity_ceil(const Quantity &q, int scale10) {
    __int128 v = q.mant;
    int e = q.exp10 + scale10;
    const __int128 lim = (__int128)INT64_MAX;
    for (int k = 0; k < q.exp2; k++) {
        v *= 2;
        if (v > lim * 1024 || v < -lim * 1024) return v > 0 ? INT64_MAX : INT64_MIN;
    }
    while (e > 0) {
        v *= 10, e--;
        if (v > lim * 1024 || v < -lim * 1024) return v > 0 ? INT64_MAX : INT64_MIN;
    }
    if (e < 0) {
        __int128 d = 1;
        bool huge = false;
        for (int k = 0; k < -e; k++) {
            d *= 10;
            if (d > ((__int128)1 << 110)) {
                huge = true;
                break;
            }
        }
        if (huge) v = v > 0 ? 1 : 0; // 0 < |x| < 1: ceil is 1 for positive, 0 for negative values
        else {
            const __int128 qd = v / d, r = v % d; // truncation toward zero
            v = qd + ((r > 0) ? 1 : 0);
        }
    }
    if (v > lim) return INT64_MAX;
    if (v < -lim) return INT64_MIN;
    return (int64_t)v;
}

// ---- Quantity.String(), for the report's podRequirements (report.go:111-144 sums Quantities and prints them) ----------------
enum class QuantityFormat { DecimalSI, BinarySI, DecimalExponent };
// the Format a parsed Quantity carries (quantity.go:283-384 -> suffix.go interpret)
inline QuantityFormat quantity_format(const std::string &text) {
    size_t e = text.size();
    while (e > 0 && (text[e - 1] == ' ' || text[e - 1] == '\t')) e--;
    if (e >= 2 && text[e - 1] == 'i' && std::string("KMGTPE").find(text[e - 2]) != std::string::npos) return QuantityFormat::BinarySI;
    for (size_t i = 0; i < e; i++)
        if ((text[i] == 'e' || text[i] == 'E') && i + 1 < e) return QuantityFormat::DecimalExponent; // (a trailing E alone is exa)
    return QuantityFormat::DecimalSI;
}
// the value in nano units, rounded up (ParseQuantity keeps nine decimal places, rounding up); non-negative quantities only
inline __int128 quantity_nano(const Quantity &q) {
    __int128 v = q.mant;
    for (int k = 0; k < q.exp2; k++) v *= 2;
    int e = q.exp10 + 9;
    while (e > 0) v *= 10, e--;
    if (e < 0) {
        __int128 d = 1;
        for (int k = 0; k < -e && d < ((__int128)1 << 120); k++) d *= 10;
        v = v / d + (v % d > 0 ? 1 : 0);
    }
    return v;
}
inline std::string int128_text(__int128 v) {
    if (v == 0) return "0";
    std::string s;
    for (; v > 0; v /= 10) s.insert(s.begin(), (char)('0' + (int)(v % 10)));
    return s;
}
// quantity.go:424-461 CanonicalizeBytes; amount.go:257-293; math.go:262-287: BinarySI prints an integer >= 1024 as <n><Ki|Mi|..> with
// every factor of 1024 removed and falls back to DecimalSI below 1024 or for a fractional value; the decimal forms print
// mantissa x 10^exponent with the trailing zeros removed and the exponent lowered to a multiple of 3 (12000 -> 12k, 1.5 -> 1500m)
inline std::string quantity_canonical(__int128 nano, QuantityFormat fmt) {
    if (nano == 0) return "0";
    const __int128 G = 1000000000;
    if (fmt == QuantityFormat::BinarySI) {
        if (nano < 1024 * G || nano % G != 0) fmt = QuantityFormat::DecimalSI;
        else {
            static const char *suf[] = {"", "Ki", "Mi", "Gi", "Ti", "Pi", "Ei"};
            __int128 m = nano / G;
            int t = 0;
            while (m >= 1024 && m % 1024 == 0 && t < 6) m /= 1024, t++;
            return int128_text(m) + suf[t];
        }
    }
    __int128 m = nano;
    int e = -9;
    while (m >= 10 && m % 10 == 0) m /= 10, e++;
    switch (e % 3) { // (C++ and Go agree: the remainder keeps the sign of the dividend)
    case 1: case -2: m *= 10, e -= 1; break;
    case 2: case -1: m *= 100, e -= 2; break;
    }
    if (fmt == QuantityFormat::DecimalExponent) return int128_text(m) + (e ? "e" + std::to_string(e) : "");
    static const char *dec[] = {"n", "u", "m", "", "k", "M", "G", "T", "P", "E"};
    return int128_text(m) + (e >= -9 && e <= 18 ? dec[(e + 9) / 3] : "");
}

// The snapshot's integers: a single quantity is refused beyond 2^60 (1Ei -- no node holds that, and Quantity itself leaves int64 just
// above), sums are formed with add64, which refuses to leave int64: never a silent wrap-around.
inline int64_t quantity_in_range(int64_t v, const std::string &text) {
    constexpr int64_t lim = (int64_t)1 << 60;
    if (v > lim || v < -lim) throw std::runtime_error("quantity '" + text + "' is out of range (beyond 2^60)");
    return v;
}
inline int64_t add64(int64_t a, int64_t b) {
    int64_t r;
    if (__builtin_add_overflow(a, b, &r)) throw std::runtime_error("resource quantities sum beyond int64");
    return r;
}
inline int64_t quantity_value(const std::string &text) { return quantity_in_range(quantity_ceil(parse_quantity(text), 0), text); }
inline int64_t quantity_milli_value(const std::string &text) { return quantity_in_range(quantity_ceil(parse_quantity(text), 3), text); }

} // namespace cchost
