// oracle/orc_opcount.h — TEST INFRASTRUCTURE ONLY (see oracle.cpp's header).
//
// A counting stand-in for `double`, used by ONE extra build of the oracle (liboracle_opcount.so, -DORC_COUNT_OPS): oracle.cpp is
// compiled with `double` naming this type, so every f64 operation the restatement executes per sample — the reference's arithmetic,
// expression by expression — is tallied by kind.  tests/sweeps/measure_ops_per_sample.py turns the tallies into the committed
// F64_OPS_PER_SAMPLE constants of raytracinginrust_amd/workloads.py, which bench.py prices against the f64 VALU issue peak.
// The arithmetic itself is unchanged (each operator performs the plain IEEE operation on the wrapped value), so this build renders
// the same samples as the normal one (checked by tests/test_oracle_opcount.py).
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <type_traits>

namespace orc_ops {

enum Kind { ADD = 0, MUL, DIV, SQRT, CMP, MINMAX, SIN, COS, TAN, ATAN, ATAN2, ACOS, LOG, LOG2, POW, FLOOR, NEGABS, CVT, N_KINDS };
struct Tally { uint64_t n[N_KINDS]; };
inline thread_local Tally tl = {};
inline void tick(Kind k) { tl.n[k]++; }

typedef double raw_f64;

struct Counted {
    raw_f64 v;
    Counted() = default;
    Counted(raw_f64 x) : v(x) {}
    Counted(float x) : v(x) {}
    template <typename I, typename = typename std::enable_if<std::is_integral<I>::value>::type>
    Counted(I x) : v((raw_f64)x) { tick(CVT); }
    template <typename I, typename = typename std::enable_if<std::is_integral<I>::value>::type>
    explicit operator I() const { tick(CVT); return (I)v; }
    explicit operator float() const { return (float)v; }
    raw_f64 raw() const { return v; }

    friend Counted operator+(Counted a, Counted b) { tick(ADD); return Counted(a.v + b.v); }
    friend Counted operator-(Counted a, Counted b) { tick(ADD); return Counted(a.v - b.v); }
    friend Counted operator*(Counted a, Counted b) { tick(MUL); return Counted(a.v * b.v); }
    friend Counted operator/(Counted a, Counted b) { tick(DIV); return Counted(a.v / b.v); }
    friend Counted operator-(Counted a) { tick(NEGABS); return Counted(-a.v); }
    friend Counted operator+(Counted a) { return a; }
    Counted& operator+=(Counted b) { tick(ADD); v += b.v; return *this; }
    Counted& operator-=(Counted b) { tick(ADD); v -= b.v; return *this; }
    Counted& operator*=(Counted b) { tick(MUL); v *= b.v; return *this; }
    Counted& operator/=(Counted b) { tick(DIV); v /= b.v; return *this; }
    friend bool operator<(Counted a, Counted b) { tick(CMP); return a.v < b.v; }
    friend bool operator>(Counted a, Counted b) { tick(CMP); return a.v > b.v; }
    friend bool operator<=(Counted a, Counted b) { tick(CMP); return a.v <= b.v; }
    friend bool operator>=(Counted a, Counted b) { tick(CMP); return a.v >= b.v; }
    friend bool operator==(Counted a, Counted b) { tick(CMP); return a.v == b.v; }
    friend bool operator!=(Counted a, Counted b) { tick(CMP); return a.v != b.v; }
};
static_assert(sizeof(Counted) == sizeof(raw_f64) && std::is_trivially_copyable<Counted>::value && std::is_standard_layout<Counted>::value,
              "Counted must have double's layout: the C API passes arrays of it to ctypes as doubles");

} // namespace orc_ops

// oracle.cpp calls the <cmath> functions with the std:: qualifier; these overloads take the counting type.
namespace std {
inline orc_ops::Counted sqrt(orc_ops::Counted x) { orc_ops::tick(orc_ops::SQRT); return orc_ops::Counted(std::sqrt(x.v)); }
inline orc_ops::Counted sin(orc_ops::Counted x) { orc_ops::tick(orc_ops::SIN); return orc_ops::Counted(std::sin(x.v)); }
inline orc_ops::Counted cos(orc_ops::Counted x) { orc_ops::tick(orc_ops::COS); return orc_ops::Counted(std::cos(x.v)); }
inline orc_ops::Counted tan(orc_ops::Counted x) { orc_ops::tick(orc_ops::TAN); return orc_ops::Counted(std::tan(x.v)); }
inline orc_ops::Counted atan(orc_ops::Counted x) { orc_ops::tick(orc_ops::ATAN); return orc_ops::Counted(std::atan(x.v)); }
inline orc_ops::Counted atan2(orc_ops::Counted y, orc_ops::Counted x) { orc_ops::tick(orc_ops::ATAN2); return orc_ops::Counted(std::atan2(y.v, x.v)); }
inline orc_ops::Counted acos(orc_ops::Counted x) { orc_ops::tick(orc_ops::ACOS); return orc_ops::Counted(std::acos(x.v)); }
inline orc_ops::Counted log(orc_ops::Counted x) { orc_ops::tick(orc_ops::LOG); return orc_ops::Counted(std::log(x.v)); }
inline orc_ops::Counted log2(orc_ops::Counted x) { orc_ops::tick(orc_ops::LOG2); return orc_ops::Counted(std::log2(x.v)); }
inline orc_ops::Counted pow(orc_ops::Counted x, orc_ops::Counted y) { orc_ops::tick(orc_ops::POW); return orc_ops::Counted(std::pow(x.v, y.v)); }
inline orc_ops::Counted floor(orc_ops::Counted x) { orc_ops::tick(orc_ops::FLOOR); return orc_ops::Counted(std::floor(x.v)); }
inline orc_ops::Counted fabs(orc_ops::Counted x) { orc_ops::tick(orc_ops::NEGABS); return orc_ops::Counted(std::fabs(x.v)); }
inline orc_ops::Counted fmax(orc_ops::Counted a, orc_ops::Counted b) { orc_ops::tick(orc_ops::MINMAX); return orc_ops::Counted(std::fmax(a.v, b.v)); }
inline orc_ops::Counted fmin(orc_ops::Counted a, orc_ops::Counted b) { orc_ops::tick(orc_ops::MINMAX); return orc_ops::Counted(std::fmin(a.v, b.v)); }
inline bool isfinite(orc_ops::Counted x) { return std::isfinite(x.v); }
template <> struct numeric_limits<orc_ops::Counted> {
    static constexpr bool is_specialized = true;
    static orc_ops::Counted max() { return orc_ops::Counted(numeric_limits<double>::max()); }
    static orc_ops::Counted infinity() { return orc_ops::Counted(numeric_limits<double>::infinity()); }
};
} // namespace std
