// kat_vm.hpp -- a small stack machine that runs the known-answer vectors of tests/golden/known_answer_vectors.json
// (the reference's own unit tests for its numeric types, re-expressed as data by tests/golden/make_known_answer_vectors.py)
// against THIS repository's arithmetic: csrc/hdr_math.hpp, bla_math.hpp, at_math.hpp.  One definition, two builds:
// kat_host.cpp (g++, the host functions) and kat_device.hip (hipcc gfx950, one kernel; PerformAT then runs the kernels' own
// tuned at_perform).  Test infrastructure: nothing in the product includes this file.
#pragma once

#include <stdint.h>

#include "../../fractalshark_amd/csrc/hdr_math.hpp"
#include "../../fractalshark_amd/csrc/bla_math.hpp"
#include "../../fractalshark_amd/csrc/at_math.hpp"

namespace kat {
using namespace fs;

// instruction kinds
enum { I_PUSH_INT = 0, I_PUSH_DBL, I_PUSH_FLT, I_LOAD, I_STORE, I_CALL, I_ASSERT_NEAR, I_ASSERT_EQ, I_ASSERT_TRUE, I_ASSERT_FALSE };

// operation vocabulary (tests/test_known_answers.py reads this list to number the names)
#define KAT_FUNCS(X)                                                                                                \
    X(DBL_INF) X(DBL_MAX) X(FLT_MAX) X(MIN_BIG_EXP) X(HDRMax) X(HDRMin) X(MakeSimpleATInfo) X(PerformAT) X(Reduce)          \
    X(ReduceGet) X(add) X(sub) X(mul) X(div) X(neg) X(eq) X(ne) X(lt) X(gt) X(le) X(ge) X(or) X(and) X(cast_double)           \
    X(cast_float) X(cast_int) X(to_float) X(chebychevNorm) X(compareTo) X(ctor_ATInfoD) X(ctor_ATResultD) X(ctor_BLAd)         \
    X(ctor_FC) X(ctor_HDRCd) X(ctor_HDRCf) X(ctor_HDRd) X(ctor_HDRf) X(divide2) X(divide2_mutable) X(divide4)                  \
    X(divide4_mutable) X(multiply2) X(multiply2_mutable) X(multiply4) X(multiply4_mutable) X(fabs) X(ldexp) X(pow)           \
    X(field_StepLength) X(field_bla_iterations) X(field_bla_steps) X(getC) X(getDZ) X(getExp) X(getIm) X(getRe) X(getL)       \
    X(getMantissa) X(getMultiplier_d) X(getMultiplier_f) X(getNewA) X(getNewB) X(getR2) X(getValue) X(hypotA) X(hypotB)      \
    X(isValid) X(negate) X(norm) X(norm_squared) X(reciprocal) X(setExp) X(square) X(toDoubleSub)
enum Func {
#define X(n) F_##n,
    KAT_FUNCS(X)
#undef X
        F_COUNT
};

enum Kind { K_INT = 0, K_DBL, K_FLT, K_R, K_RF, K_C, K_CF, K_BLA, K_AT, K_RES };

struct AtInfo { // the fields of ATInfo the hot path uses (ATInfo.h:60-106)
    int64_t StepLength;
    hreal<double> SqrEscapeRadius, ThresholdC;
    hcplx<double> RefC, ZCoeff, CCoeff, InvZCoeff;
};
struct AtRes {
    int64_t bla_iterations, bla_steps;
};

struct Val {
    int kind;
    int64_t i;
    double d;
    float f;
    hreal<double> r;
    hreal<float> rf;
    hcplx<double> c;
    hcplx<float> cf;
};

struct Instr {
    int32_t op;
    int32_t a;  // slot / function id
    int32_t n;  // argument count of a call
    int32_t pad;
    double d;   // literal
};

struct Check { // one evaluated assertion
    int32_t ok;
    int32_t kind; // I_ASSERT_*
    double got, want, tol;
};

FS_HD Val v_int(int64_t i) { Val v{}; v.kind = K_INT; v.i = i; return v; }
FS_HD Val v_dbl(double d) { Val v{}; v.kind = K_DBL; v.d = d; return v; }
FS_HD Val v_flt(float f) { Val v{}; v.kind = K_FLT; v.f = f; return v; }
FS_HD Val v_r(hreal<double> r) { Val v{}; v.kind = K_R; v.r = r; return v; }
FS_HD Val v_rf(hreal<float> r) { Val v{}; v.kind = K_RF; v.rf = r; return v; }
FS_HD Val v_c(hcplx<double> c) { Val v{}; v.kind = K_C; v.c = c; return v; }
FS_HD Val v_cf(hcplx<float> c) { Val v{}; v.kind = K_CF; v.cf = c; return v; }

// numeric value of a scalar-like operand (HDRFloat -> its toDouble(), as static_cast<double> does)
FS_HD double num(const Val &v)
{
    switch (v.kind) {
        case K_INT: return (double)v.i;
        case K_DBL: return v.d;
        case K_FLT: return (double)v.f;
        case K_R: return hr_to_native<double>(v.r);
        case K_RF: return (double)hr_to_native<float>(v.rf);
        default: return 0.0 / 0.0;
    }
}
FS_HD bool is_num(const Val &v) { return v.kind == K_INT || v.kind == K_DBL || v.kind == K_FLT; }

struct Machine {
    // big objects live beside the value stack: a Val refers to them by index
    BlaRec<double> bla[80];
    AtInfo at[4];
    AtRes res[8];
    int n_bla = 0, n_at = 0, n_res = 0;
    Val stack[24];
    int sp = 0;
    Val slots[80];

    FS_HD void push(const Val &v) { stack[sp++] = v; }
    FS_HD Val pop() { return stack[--sp]; }
    FS_HD Val obj(int kind, int idx) { Val v{}; v.kind = kind; v.i = idx; return v; }

    FS_HD Val arith(int f, const Val &a, const Val &b)
    {
        if (a.kind == K_INT && b.kind == K_INT) {
            switch (f) {
                case F_add: return v_int(a.i + b.i);
                case F_sub: return v_int(a.i - b.i);
                case F_mul: return v_int(a.i * b.i);
                default: return v_int(b.i ? a.i / b.i : 0);
            }
        }
        if (a.kind == K_FLT && b.kind == K_FLT) {
            switch (f) {
                case F_add: return v_flt(a.f + b.f);
                case F_sub: return v_flt(a.f - b.f);
                case F_mul: return v_flt(a.f * b.f);
                default: return v_flt(a.f / b.f);
            }
        }
        if (is_num(a) && is_num(b)) {
            const double x = num(a), y = num(b);
            switch (f) {
                case F_add: return v_dbl(x + y);
                case F_sub: return v_dbl(x - y);
                case F_mul: return v_dbl(x * y);
                default: return v_dbl(x / y);
            }
        }
        if (a.kind == K_R && b.kind == K_R) {
            switch (f) {
                case F_add: return v_r(hr_add(a.r, b.r));
                case F_sub: return v_r(hr_sub(a.r, b.r));
                case F_mul: return v_r(hr_mul(a.r, b.r));
                default: return v_r(hr_div(a.r, b.r));
            }
        }
        if (a.kind == K_RF && b.kind == K_RF) {
            switch (f) {
                case F_add: return v_rf(hr_add(a.rf, b.rf));
                case F_sub: return v_rf(hr_sub(a.rf, b.rf));
                case F_mul: return v_rf(hr_mul(a.rf, b.rf));
                default: return v_rf(hr_div(a.rf, b.rf));
            }
        }
        if (a.kind == K_C && b.kind == K_C) {
            switch (f) {
                case F_add: return v_c(hc_add(a.c, b.c));
                case F_sub: return v_c(hc_add(a.c, hcplx<double>{-b.c.re, -b.c.im, b.c.e})); // sub_mutable = plus of the negation
                case F_mul: return v_c(hc_mul(a.c, b.c));
                default: return v_c(hc_div(a.c, b.c));
            }
        }
        if (a.kind == K_C && (b.kind == K_R || is_num(b))) { // complex (+ | *) real; a plain scalar enters as HDRFloat(T mant)
            const hreal<double> r = b.kind == K_R ? b.r : to_r(b);
            if (f == F_add)
                return v_c(hc_add_real(a.c, r));
            if (f == F_mul)
                return v_c(hc_mul_real(a.c, r));
        }
        return v_dbl(0.0 / 0.0);
    }

    FS_HD hreal<double> to_r(const Val &v)
    {
        if (v.kind == K_R)
            return v.r;
        if (v.kind == K_INT)
            return hr_from_number<double>((double)v.i); // templated scalar constructor: zero carries the minimum exponent
        return hr_from_mant<double>(num(v));            // HDRFloat(T mant)
    }

    FS_HD void call(int f, int n)
    {
        Val a[6];
        for (int k = n - 1; k >= 0; k--)
            a[k] = pop();
        switch (f) {
            case F_DBL_INF: push(v_dbl(1.0 / 0.0)); break;
            case F_DBL_MAX: push(v_dbl(type_max<double>())); break;
            case F_FLT_MAX: push(v_flt(type_max<float>())); break;
            case F_MIN_BIG_EXP: push(v_int(kMinBigExp)); break;
            case F_add: case F_sub: case F_mul: case F_div: push(arith(f, a[0], a[1])); break;
            case F_neg:
                if (a[0].kind == K_INT) push(v_int(-a[0].i));
                else if (a[0].kind == K_FLT) push(v_flt(-a[0].f));
                else push(v_dbl(-num(a[0])));
                break;
            case F_eq: case F_ne: {
                bool e;
                if (a[0].kind == K_R && a[1].kind == K_R)
                    e = a[0].r.m == a[1].r.m && a[0].r.e == a[1].r.e; // operator== compares the raw fields
                else if (a[0].kind == K_C && a[1].kind == K_C)
                    e = a[0].c.re == a[1].c.re && a[0].c.im == a[1].c.im && a[0].c.e == a[1].c.e;
                else
                    e = num(a[0]) == num(a[1]);
                push(v_int(f == F_eq ? e : !e));
                break;
            }
            case F_lt: push(v_int(num(a[0]) < num(a[1]))); break;
            case F_gt: push(v_int(num(a[0]) > num(a[1]))); break;
            case F_le: push(v_int(num(a[0]) <= num(a[1]))); break;
            case F_ge: push(v_int(num(a[0]) >= num(a[1]))); break;
            case F_or: push(v_int(num(a[0]) != 0.0 || num(a[1]) != 0.0)); break;
            case F_and: push(v_int(num(a[0]) != 0.0 && num(a[1]) != 0.0)); break;
            case F_cast_double: push(v_dbl(num(a[0]))); break;
            case F_cast_float: case F_to_float: push(v_flt((float)num(a[0]))); break;
            case F_cast_int: push(v_int((int64_t)num(a[0]))); break;
            case F_fabs: push(v_dbl(num(a[0]) < 0 ? -num(a[0]) : num(a[0]))); break;
            case F_ldexp: push(v_dbl(num(a[0]) * multiplier<double>((int32_t)a[1].i))); break;
            case F_pow: { // (integer powers of two only: std::pow(2.0, 32))
                double r = 1.0;
                for (int64_t k = 0; k < (int64_t)num(a[1]); k++)
                    r *= num(a[0]);
                push(v_dbl(r));
                break;
            }
            // ---- HDRFloat
            case F_ctor_HDRd:
                if (n == 0) push(v_r(hr_zero<double>()));
                else if (n == 1) push(v_r(to_r(a[0])));
                else push(v_r(hr_raw<double>((int32_t)a[0].i, num(a[1]))));
                break;
            case F_ctor_HDRf:
                if (n == 0) push(v_rf(hr_zero<float>()));
                else if (n == 1) push(v_rf(a[0].kind == K_INT ? hr_from_number<float>((float)a[0].i) : hr_from_mant<float>((float)num(a[0]))));
                else push(v_rf(hr_raw<float>((int32_t)a[0].i, (float)num(a[1]))));
                break;
            case F_getMantissa: push(a[0].kind == K_RF ? v_flt(a[0].rf.m) : v_dbl(a[0].r.m)); break;
            case F_getExp: push(v_int(a[0].kind == K_RF ? a[0].rf.e : a[0].r.e)); break;
            case F_Reduce: push(a[0].kind == K_RF ? v_rf(hr_reduced(a[0].rf)) : v_r(hr_reduced(a[0].r))); break;
            case F_ReduceGet: {
                const hreal<double> r = hr_reduced(a[0].r);
                push(v_r(r));
                push(v_int(r.e - a[0].r.e));
                break;
            }
            case F_setExp: { Val v = a[0]; if (v.kind == K_RF) v.rf.e = (int32_t)a[1].i; else v.r.e = (int32_t)a[1].i; push(v); break; }
            case F_negate: push(v_r(hr_neg(a[0].r))); break;
            case F_square: push(a[0].kind == K_RF ? v_rf(hr_square(a[0].rf)) : v_r(hr_square(a[0].r))); break;
            case F_multiply2: case F_multiply2_mutable: push(v_r(hr_mul2(a[0].r))); break;
            case F_multiply4: case F_multiply4_mutable: push(v_r(hreal<double>{a[0].r.m, a[0].r.e + 2})); break;
            case F_divide2: push(v_r(hreal<double>{a[0].r.m, a[0].r.e - 1})); break;
            case F_divide2_mutable: push(v_r(hreal<double>{a[0].r.m, clamp_exp(a[0].r.e - 1)})); break;
            case F_divide4: push(v_r(hreal<double>{a[0].r.m, a[0].r.e - 2})); break;
            case F_divide4_mutable: push(v_r(hreal<double>{a[0].r.m, clamp_exp(a[0].r.e - 2)})); break;
            case F_compareTo: push(v_int(hr_cmp(a[0].r, a[1].r))); break;
            case F_HDRMax: push(v_r(hr_cmp(a[0].r, a[1].r) > 0 ? a[0].r : a[1].r)); break;
            case F_HDRMin: push(v_r(hr_cmp(a[0].r, a[1].r) < 0 ? a[0].r : a[1].r)); break;
            case F_toDoubleSub: push(v_dbl(a[0].r.m * multiplier<double>(a[0].r.e - (int32_t)a[1].i))); break;
            case F_getMultiplier_d: push(v_dbl(multiplier<double>((int32_t)a[0].i))); break;
            case F_getMultiplier_f: push(v_flt(multiplier<float>((int32_t)a[0].i))); break;
            case F_reciprocal:
                if (a[0].kind == K_C) push(v_c(hc_recip(a[0].c)));
                else push(v_r(hr_recip(a[0].r)));
                break;
            // ---- HDRFloatComplex (FloatComplex<double> operands are carried as reduced HDR complex values)
            case F_ctor_HDRCd: case F_ctor_FC:
                if (n == 0) push(v_c(hc_zero<double>()));
                else if (n == 1 && a[0].kind == K_C) push(a[0]); // copy construction
                else if (n == 1) push(v_c(hcplx<double>{(double)a[0].cf.re, (double)a[0].cf.im, a[0].cf.e}));
                else if (a[0].kind == K_R) push(v_c(hc_from_hr(a[0].r, a[1].r)));
                else push(v_c(hc_from_native<double>(num(a[0]), num(a[1]))));
                break;
            case F_ctor_HDRCf: push(v_cf(hc_from_native<float>((float)num(a[0]), (float)num(a[1])))); break;
            case F_getRe: push(v_r(hc_re(a[0].c))); break;
            case F_getIm: push(v_r(hc_im(a[0].c))); break;
            case F_norm: push(v_r(hc_norm(a[0].c))); break;
            case F_norm_squared: push(v_r(hc_norm2(a[0].c))); break;
            case F_chebychevNorm: push(v_r(hc_cheb(a[0].c))); break;
            // ---- BLA<double>
            case F_ctor_BLAd: {
                BlaRec<double> b{};
                if (n == 6) {
                    b.r2 = to_r(v_dbl(num(a[0])));
                    b.Ax = to_r(v_dbl(num(a[1]))), b.Ay = to_r(v_dbl(num(a[2])));
                    b.Bx = to_r(v_dbl(num(a[3]))), b.By = to_r(v_dbl(num(a[4])));
                    b.l = (int32_t)num(a[5]);
                }
                bla[n_bla] = b;
                push(obj(K_BLA, n_bla++));
                break;
            }
            case F_getR2: push(v_r(bla[a[0].i].r2)); break;
            case F_getL: push(v_int(bla[a[0].i].l)); break;
            case F_hypotA: push(v_r(bla_hypot(bla[a[0].i].Ax, bla[a[0].i].Ay))); break;
            case F_hypotB: push(v_r(bla_hypot(bla[a[0].i].Bx, bla[a[0].i].By))); break;
            case F_getValue: {
                hreal<double> x = to_r(v_dbl(num(a[1]))), y = to_r(v_dbl(num(a[2])));
                bla_get_value(bla[a[0].i], x, y, to_r(v_dbl(num(a[3]))), to_r(v_dbl(num(a[4]))));
                push(v_dbl(hr_to_native<double>(x)));
                push(v_dbl(hr_to_native<double>(y)));
                break;
            }
            case F_getNewA: case F_getNewB: {
                hreal<double> x, y;
                if (f == F_getNewA) bla_new_a(bla[a[0].i], bla[a[1].i], x, y);
                else bla_new_b(bla[a[0].i], bla[a[1].i], x, y);
                push(v_dbl(hr_to_native<double>(x)));
                push(v_dbl(hr_to_native<double>(y)));
                break;
            }
            // ---- ATInfo
            case F_ctor_ATInfoD: at[n_at] = AtInfo{}; push(obj(K_AT, n_at++)); break;
            case F_ctor_ATResultD: res[n_res] = AtRes{}; push(obj(K_RES, n_res++)); break;
            case F_MakeSimpleATInfo: { // the test file's helper: the derived InvZCoeff = 1 / ZCoeff in plain doubles
                AtInfo t{};
                t.SqrEscapeRadius = to_r(v_dbl(num(a[0])));
                t.ThresholdC = to_r(v_dbl(num(a[1])));
                t.StepLength = (int64_t)num(a[2]);
                t.RefC = a[3].c, t.ZCoeff = a[4].c, t.CCoeff = a[5].c;
                const double zr = hr_to_native<double>(hc_re(a[4].c)), zi = hr_to_native<double>(hc_im(a[4].c));
                const double den = zr * zr + zi * zi;
                t.InvZCoeff = den > 0 ? hc_from_native<double>(zr / den, -zi / den) : hc_from_native<double>(0.0, 0.0);
                at[n_at] = t;
                push(obj(K_AT, n_at++));
                break;
            }
            case F_field_StepLength: push(v_int(at[a[0].i].StepLength)); break;
            case F_field_bla_iterations: push(v_int(res[a[0].i].bla_iterations)); break;
            case F_field_bla_steps: push(v_int(res[a[0].i].bla_steps)); break;
            case F_isValid: push(v_int(hr_cmp_pos(hc_cheb(a[1].c), at[a[0].i].ThresholdC) <= 0)); break; // ATInfo.h:128-135
            case F_getC: push(v_c(hc_reduced(hc_add(hc_mul(a[1].c, at[a[0].i].CCoeff), at[a[0].i].RefC)))); break;
            case F_getDZ: push(v_c(hc_reduced(hc_mul(a[1].c, at[a[0].i].InvZCoeff)))); break;
            case F_PerformAT: { // ATInfo.h:155-188
                const AtInfo &t = at[a[0].i];
                const uint64_t ATMaxIt = (uint64_t)a[1].i / (uint64_t)t.StepLength;
                const hcplx<double> c = hc_reduced(hc_add(hc_mul(a[2].c, t.CCoeff), t.RefC));
                hcplx<double> z;
                uint64_t i;
#if defined(__HIP_DEVICE_COMPILE__)
                at_perform<double, uint64_t>(c, t.SqrEscapeRadius, ATMaxIt, z, i); // the kernels' own (tuned) function
#else
                at_perform_literal<double, uint64_t>(c, t.SqrEscapeRadius, ATMaxIt, z, i);
#endif
                res[n_res] = AtRes{(int64_t)(i * (uint64_t)t.StepLength), (int64_t)i};
                push(obj(K_RES, n_res++));
                break;
            }
            default: push(v_dbl(0.0 / 0.0)); break;
        }
    }

    // -> number of assertions evaluated
    FS_HD int run(const Instr *prog, int n, Check *out, int max_checks)
    {
        int nc = 0;
        for (int pc = 0; pc < n; pc++) {
            const Instr &in = prog[pc];
            switch (in.op) {
                case I_PUSH_INT: push(v_int((int64_t)in.d)); break;
                case I_PUSH_DBL: push(v_dbl(in.d)); break;
                case I_PUSH_FLT: push(v_flt((float)in.d)); break;
                case I_LOAD: push(slots[in.a]); break;
                case I_STORE: slots[in.a] = pop(); break;
                case I_CALL: call(in.a, in.n); break;
                default: {
                    Check c{};
                    c.kind = in.op;
                    if (in.op == I_ASSERT_NEAR) {
                        const Val tol = pop(), want = pop(), got = pop();
                        c.got = num(got), c.want = num(want), c.tol = num(tol);
                        const double d = c.got - c.want;
                        c.ok = (d < 0 ? -d : d) <= c.tol;
                    } else if (in.op == I_ASSERT_EQ) {
                        const Val want = pop(), got = pop();
                        c.got = num(got), c.want = num(want);
                        c.ok = c.got == c.want;
                    } else {
                        const Val got = pop();
                        c.got = num(got), c.want = in.op == I_ASSERT_TRUE ? 1.0 : 0.0;
                        c.ok = (c.got != 0.0) == (in.op == I_ASSERT_TRUE);
                    }
                    if (nc < max_checks)
                        out[nc] = c;
                    nc++;
                }
            }
        }
        return nc;
    }
};

} // namespace kat
