// oracle/df_ref.hpp -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see gpu_ref_2x32.cpp's header).
//
// CPU restatement of the reference's double-float ("2x32") primitives, shared by the 2x32 oracles:
//   dblflt primitives        HpSharkFloatLib/dblflt.cuh:86-215 (add_float_to_dblflt, add/sub/mul_dblflt), dblflt.h:19-29
//   CudaDblflt compare/abs   HpSharkFloatLib/CudaDblflt.h:197-259
// Every __fadd_rn / __fmul_rn / __fmaf_rn is one correctly rounded IEEE binary32 operation (build: -ffp-contract=off,
// fmaf() = libm's correctly rounded FMA).  Nothing in the product path may include this file.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace {

inline uint32_t f2u(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
inline float u2f(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// ------------------------------------------------------------------ dblflt (MattDblflt + dblflt.cuh)
struct DF {
    float head;
    float tail;
};

// MattDblflt(float a, float b), dblflt.h:19-29 (== add_float_to_dblflt, dblflt.cuh:86-97)
inline DF DFTwoSum(float a, float b)
{
    DF z;
    z.head = a + b;
    float t1 = z.head - a;
    float t2 = z.head - t1;
    t1 = b - t1;
    t2 = a - t2;
    z.tail = t1 + t2;
    return z;
}
// CudaDblflt(float) -> MattDblflt(float) -> MattDblflt{other, 0.0f}, CudaDblflt.h:64-65, dblflt.h:54-55
inline DF DFFromFloat(float f) { return DFTwoSum(f, 0.0f); }
inline DF DFZero() { return DF{0.0f, 0.0f}; } // CudaDblflt(), CudaDblflt.h:41-42
inline DF DFNeg(DF a) { return DF{-a.head, -a.tail}; } // CudaDblflt.h:181-186

// add_dblflt, dblflt.cuh:116-132
inline DF DFAdd(DF a, DF b)
{
    float t1 = a.head + b.head;
    float t2 = t1 + -a.head;
    float t3 = (a.head + (t2 - t1)) + (b.head + -t2);
    float t4 = a.tail + b.tail;
    t2 = t4 + -a.tail;
    float t5 = (a.tail + (t2 - t4)) + (b.tail + -t2);
    t3 = t3 + t4;
    t4 = t1 + t3;
    t3 = (t1 - t4) + t3;
    t3 = t3 + t5;
    DF z;
    const float e = t4 + t3;
    z.head = e;
    z.tail = (t4 - e) + t3;
    return z;
}
// sub_dblflt, dblflt.cuh:141-157
inline DF DFSub(DF a, DF b)
{
    float t1 = a.head + -b.head;
    float t2 = t1 + -a.head;
    float t3 = (a.head + (t2 - t1)) + -(b.head + t2);
    float t4 = a.tail + -b.tail;
    t2 = t4 + -a.tail;
    float t5 = (a.tail + (t2 - t4)) + -(b.tail + t2);
    t3 = t3 + t4;
    t4 = t1 + t3;
    t3 = (t1 - t4) + t3;
    t3 = t3 + t5;
    DF z;
    const float e = t4 + t3;
    z.head = e;
    z.tail = (t4 - e) + t3;
    return z;
}
// mul_dblflt, dblflt.cuh:164-176
inline DF DFMul(DF a, DF b)
{
    DF t;
    t.head = a.head * b.head;
    t.tail = fmaf(a.head, b.head, -t.head);
    t.tail = fmaf(a.tail, b.tail, t.tail);
    t.tail = fmaf(a.head, b.tail, t.tail);
    t.tail = fmaf(a.tail, b.head, t.tail);
    DF z;
    const float e = t.head + t.tail;
    z.head = e;
    z.tail = (t.head - e) + t.tail;
    return z;
}
// CudaDblflt comparisons, CudaDblflt.h:197-245
inline bool DFLt(DF a, DF b) { return a.head < b.head || (a.head == b.head && a.tail < b.tail); }
inline bool DFEq(DF a, DF b) { return a.head == b.head && a.tail == b.tail; }
inline bool DFGt(DF a, DF b) { return !DFLt(a, b) && !DFEq(b, a); }
inline bool DFGe(DF a, DF b) { return !DFLt(a, b); }
// CudaDblflt::abs, CudaDblflt.h:247-255
inline DF DFAbs(DF a) { return DFLt(a, DFFromFloat(0.0f)) ? DFNeg(a) : a; }

} // namespace
