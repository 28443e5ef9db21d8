#!/usr/bin/env python3
"""Known-answer vectors of the reference's own unit tests for the numeric core -> tests/golden/known_answer_vectors.json.

Source (read where it lies, /root/reference; nothing of it is copied): FractalSharkTest/TestHDRFloat.cpp,
TestHDRFloatComplex.cpp, TestATInfo.cpp, TestBLA.cpp.  Each TEST(...) there is a short straight-line program over the
reference's numeric types that ends in ASSERT_NEAR / ASSERT_EQ / ASSERT_TRUE checks.  This script reads those programs and
re-expresses every one as DATA: a list of stack-machine operations (push this operand, call that operation, store, compare
with this expected value under this tolerance) over an operation vocabulary of this repository's own (tests/kat/kat_vm.hpp maps
each operation name onto csrc/hdr_math.hpp / bla_math.hpp / at_math.hpp).  What is committed is the operands, the operation
names, the expected values and the tolerances -- the vectors -- not the text of the tests.

Loops with constant bounds are unrolled, constants folded.  Tests that exercise things outside the hot path's numeric core
(MPIR HighPrecision conversion, text I/O, the alternative member order, 64-bit exponents) are listed as skipped with the
reason.  Usage: python tests/golden/make_known_answer_vectors.py   (needs /root/reference)."""
import json
import os
import re
import sys

REF = "/root/reference/FractalSharkTest"
FILES = ["TestHDRFloat.cpp", "TestHDRFloatComplex.cpp", "TestATInfo.cpp", "TestBLA.cpp", "TestFloatComplex.cpp"]
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "known_answer_vectors.json")

SKIP_IF = [("ostringstream", "text I/O of FloatComplex: not on the hot path"), ("HighPrecision", "MPIR HighPrecision conversion: host-only, outside the per-pixel numeric core"),
           ("ToString", "text I/O of HDRFloat: not on the hot path"),
           ("HDROrder::Right", "alternative member order of the reference's template: this repository has one layout"),
           ("HRReal", "64-bit exponent instantiation (Imagina::HRReal): the device types carry int32 exponents")]

TYPES = {"HDRd", "HDRf", "HDRCd", "HDRCf", "double", "float", "int", "int32_t", "BLAd", "ATInfoD", "ATResultD", "FC", "auto",
         "uint64_t"}


def preprocess(body):
    body = re.sub(r"//[^\n]*", "", body)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    subs = [(r"static_cast<double>\(", "cast_double("), (r"static_cast<float>\(", "cast_float("),
            (r"static_cast<uint64_t>\(", "cast_int("), (r"static_cast<int64_t>\(", "cast_int("),
            (r"std::numeric_limits<double>::infinity\(\)", "DBL_INF"), (r"std::numeric_limits<double>::max\(\)", "DBL_MAX"),
            (r"std::numeric_limits<float>::max\(\)", "FLT_MAX"), (r"std::ldexp", "ldexp"), (r"std::abs", "fabs"),
            (r"std::pow", "pow"), (r"HDRd::MIN_BIG_EXPONENT\(\)", "MIN_BIG_EXP"), (r"HDRd::getMultiplier", "getMultiplier_d"),
            (r"HDRf::getMultiplier", "getMultiplier_f"), (r"HDRd::HDRMax", "HDRMax"), (r"HDRd::HDRMin", "HDRMin"),
            (r"BLAd::getNewA", "getNewA"), (r"BLAd::getNewB", "getNewB"), (r"BLAd::getGenericStep", "BLAd"),
            (r"\.Reduce<true>\(&(\w+)\)", r".ReduceGet(\1)"), (r"HDRFloatComplex<float>", "HDRCf"),
            (r"FloatComplex<float>", "HDRCf"), (r"\bFCd\b", "FC")]
    for a, b in subs:
        body = re.sub(a, b, body)
    return body


TOK = re.compile(r"\s*(?:(0x[0-9a-fA-F.]+p[-+]?\d+|\d+\.\d*(?:[eE][-+]?\d+)?[fF]?|\.\d+(?:[eE][-+]?\d+)?[fF]?|\d+[eE][-+]?\d+[fF]?|\d+[fF]?)"
                 r"|([A-Za-z_]\w*)|(\+\+|--|==|!=|<=|>=|&&|\|\||[-+*/%<>=!(){}\[\];,.&]))")


def tokenize(s):
    out, i = [], 0
    while i < len(s):
        m = TOK.match(s, i)
        if not m:
            if s[i:].strip() == "":
                break
            raise SyntaxError("cannot tokenize at: %r" % s[i:i + 30])
        i = m.end()
        if m.group(1):
            t = m.group(1)
            isf = t[-1] in "fF" and not t.startswith("0x")
            if isf:
                t = t[:-1]
            if re.fullmatch(r"\d+", t) and not isf:
                out.append(("int", int(t)))
            else:
                out.append(("flt" if isf else "dbl", float(t) if not t.startswith("0x") else float.fromhex(t)))
        elif m.group(2):
            out.append(("id", m.group(2)))
        else:
            out.append(("op", m.group(3)))
    return out


class Unsupported(Exception):
    pass


class Compiler:
    """Statement / expression compiler: C-like test body -> list of stack-machine operations."""

    def __init__(self, toks):
        self.t, self.i = toks, 0
        self.ops = []
        self.consts = {}    # compile-time integer constants (loop counters, `const int`)
        self.vars = {}      # name -> slot
        self.vtype = {}     # name -> declared type (for method dispatch on mutable members)

    # ---- token helpers
    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", None)

    def take(self, kind=None, val=None):
        tok = self.peek()
        if (kind and tok[0] != kind) or (val is not None and tok[1] != val):
            raise SyntaxError("expected %s %s, got %s" % (kind, val, tok))
        self.i += 1
        return tok

    def at(self, val):
        return self.peek()[1] == val and self.peek()[0] == "op"

    def emit(self, *op):
        self.ops.append(list(op))

    def slot(self, name):
        if name not in self.vars:
            self.vars[name] = len(self.vars)
        return self.vars[name]

    # ---- statements
    def block(self, until="}"):
        while not (self.peek()[0] == "eof" or self.at(until)):
            self.statement()

    def statement(self):
        tok = self.peek()
        if self.at("{"):
            self.take()
            self.block()
            self.take("op", "}")
            return
        if self.at(";"):
            self.take()
            return
        if tok == ("id", "for"):
            return self.for_loop()
        is_const = tok == ("id", "const")
        if is_const:
            self.take()
            tok = self.peek()
        if tok[0] == "id" and tok[1] in TYPES and self.peek(1)[0] == "id":
            return self.declaration(is_const)
        self.expr_statement()
        self.take("op", ";")

    def for_loop(self):
        self.take("id", "for")
        self.take("op", "(")
        self.take("id", "int")
        var = self.take("id")[1]
        self.take("op", "=")
        lo = self.const_expr()
        self.take("op", ";")
        assert self.take("id")[1] == var
        self.take("op", "<")
        hi = self.const_expr()
        self.take("op", ";")
        if self.at("++"):
            self.take()
            assert self.take("id")[1] == var
        else:
            assert self.take("id")[1] == var
            self.take("op", "++")
        self.take("op", ")")
        start = self.i
        for n in range(lo, hi):
            self.i = start
            self.consts[var] = n
            self.statement()
        if hi <= lo:  # skip the body once
            depth = 0
            while True:
                tok = self.take()
                depth += tok == ("op", "{")
                depth -= tok == ("op", "}")
                if depth == 0 and tok[1] in ("}", ";"):
                    break
        del self.consts[var]

    def const_expr(self):
        """Integer constant expression (loop bounds, array sizes)."""
        start = self.i
        depth = 0
        while not (depth == 0 and (self.at(";") or self.at(")") or self.at("]") or self.at(","))):
            depth += self.at("(")
            depth -= self.at(")")
            self.i += 1
        toks = self.t[start:self.i]
        s = ""
        for k, v in toks:
            s += str(self.consts[v]) if k == "id" else str(v)
        return int(eval(s, {"__builtins__": {}}))

    def declaration(self, is_const=False):
        typ = self.take("id")[1]
        while True:
            name = self.take("id")[1]
            if self.at("["):   # array: elements are separate variables name[k]
                self.take()
                self.const_expr()
                self.take("op", "]")
            elif self.at("("):  # constructor call
                self.take()
                n = self.args(")")
                self.construct(typ, n)
                self.store(name, typ)
            elif self.at("="):
                self.take()
                if is_const and typ in ("int", "int32_t") and self.all_const_until(";,"):
                    self.consts[name] = self.const_expr()
                    self.emit("push_int", self.consts[name])
                    self.store(name, typ)
                else:
                    self.expression()
                    if typ in ("float",):
                        self.emit("call", "to_float", 1)
                    self.store(name, typ)
            else:
                if typ in ("HDRd", "HDRf", "HDRCd", "HDRCf", "BLAd", "ATInfoD", "ATResultD", "FC"):
                    self.construct(typ, 0)
                    self.store(name, typ)
                else:
                    self.vtype[name] = typ
            if self.at(","):
                self.take()
                continue
            self.take("op", ";")
            return

    def all_const_until(self, stops):
        j, depth = self.i, 0
        while j < len(self.t):
            k, v = self.t[j]
            if depth == 0 and k == "op" and v in stops:
                return True
            if k == "id" and v not in self.consts:
                return False
            if k in ("dbl", "flt"):
                return False
            depth += (k, v) == ("op", "(")
            depth -= (k, v) == ("op", ")")
            j += 1
        return False

    def construct(self, typ, n):
        if typ not in ("HDRd", "HDRf", "HDRCd", "HDRCf", "BLAd", "FC", "ATInfoD", "ATResultD"):
            raise Unsupported("constructor of " + typ)
        self.emit("call", "ctor_" + typ, n)

    def store(self, name, typ=None):
        if typ and typ != "auto":
            self.vtype[name] = typ
        self.emit("store", self.slot(name))

    def lvalue_name(self):
        name = self.take("id")[1]
        if self.at("["):
            self.take()
            k = self.const_expr()
            self.take("op", "]")
            name = "%s[%d]" % (name, k)
        return name

    def expr_statement(self):
        # assignment to a plain / indexed variable?
        save = self.i
        if self.peek()[0] == "id":
            try:
                name = self.lvalue_name()
                if self.at("="):
                    self.take()
                    self.expression()
                    self.store(name)
                    return
                # compound assignment  a op= b  ==  a = a op b
                if self.peek()[0] == "op" and self.peek()[1] in ("+", "-", "*", "/") and self.peek(1) == ("op", "="):
                    op = self.take()[1]
                    self.take("op", "=")
                    self.emit("load", self.vars[name])
                    self.expression()
                    self.emit("call", self.NAMES[op], 2)
                    self.store(name)
                    return
            except (SyntaxError, KeyError):
                pass
            self.i = save
        self.expression(statement=True)

    # ---- expressions (precedence climbing)
    PREC = {"||": 1, "&&": 2, "==": 3, "!=": 3, "<": 4, ">": 4, "<=": 4, ">=": 4, "+": 5, "-": 5, "*": 6, "/": 6}
    NAMES = {"||": "or", "&&": "and", "==": "eq", "!=": "ne", "<": "lt", ">": "gt", "<=": "le", ">=": "ge", "+": "add", "-": "sub",
             "*": "mul", "/": "div"}

    def args(self, close):
        n = 0
        while not self.at(close):
            self.expression()
            n += 1
            if self.at(","):
                self.take()
        self.take("op", close)
        return n

    def expression(self, prec=1, statement=False):
        self.unary(statement)
        while self.peek()[0] == "op" and self.peek()[1] in self.PREC and self.PREC[self.peek()[1]] >= prec:
            op = self.take()[1]
            self.expression(self.PREC[op] + 1)
            self.emit("call", self.NAMES[op], 2)

    def unary(self, statement=False):
        if self.at("-"):
            self.take()
            self.unary()
            self.emit("call", "neg", 1)
            return
        if self.at("("):
            self.take()
            self.expression()
            self.take("op", ")")
            return self.postfix(None, statement)
        tok = self.take()
        if tok[0] == "int":
            self.emit("push_int", tok[1])
        elif tok[0] == "dbl":
            self.emit("push_dbl", tok[1])
        elif tok[0] == "flt":
            self.emit("push_flt", tok[1])
        elif tok[0] == "id":
            name = tok[1]
            if self.at("("):  # function / constructor / macro call
                self.take()
                if name in ("getNewA", "getNewB"):
                    self.expression()
                    self.take("op", ",")
                    self.expression()
                    self.take("op", ",")
                    a = self.lvalue_name()
                    self.take("op", ",")
                    b = self.lvalue_name()
                    self.take("op", ")")
                    self.emit("call", name, 2)
                    self.store(b)
                    self.store(a)
                    return
                if name == "ASSERT_NEAR":
                    n = self.args(")")
                    assert n == 3
                    self.emit("assert_near")
                    return
                if name in ("ASSERT_EQ", "ASSERT_TRUE", "ASSERT_FALSE"):
                    self.args(")")
                    self.emit(name.lower())
                    return
                n = self.args(")")
                if name in ("HDRd", "HDRf", "HDRCd", "HDRCf", "BLAd", "FC"):
                    self.construct(name, n)
                else:
                    self.emit("call", name, n)
            elif name in self.consts:
                self.emit("push_int", self.consts[name])
            elif name in ("DBL_INF", "DBL_MAX", "FLT_MAX", "MIN_BIG_EXP"):
                self.emit("call", name, 0)
            else:
                if self.at("["):
                    self.take()
                    k = self.const_expr()
                    self.take("op", "]")
                    name = "%s[%d]" % (name, k)
                if name not in self.vars:
                    raise Unsupported("use of undeclared name " + name)
                self.emit("load", self.vars[name])
                return self.postfix(name, statement)
        else:
            raise SyntaxError("unexpected token %s" % (tok,))
        self.postfix(None, statement)

    MUTATORS = {"Reduce", "setExp", "multiply2_mutable", "divide2_mutable", "divide4_mutable", "multiply4_mutable"}

    def postfix(self, base_name, statement):
        while self.at("."):
            self.take()
            member = self.take("id")[1]
            if not self.at("("):  # field
                self.emit("call", "field_" + member, 1)
                base_name = None
                continue
            self.take("op", "(")
            if member == "ReduceGet":
                out = self.lvalue_name()
                self.take("op", ")")
                self.emit("call", "ReduceGet", 1)
                self.store(out, "int")
                self.store(base_name)
                return
            if member == "getValue":
                a = self.lvalue_name()
                self.take("op", ",")
                b = self.lvalue_name()
                self.take("op", ",")
                self.emit("load", self.vars[a])
                self.emit("load", self.vars[b])
                self.expression()
                self.take("op", ",")
                self.expression()
                self.take("op", ")")
                self.emit("call", "getValue", 5)
                self.store(b)
                self.store(a)
                return
            if member == "PerformAT":
                self.expression()
                self.take("op", ",")
                self.expression()
                self.take("op", ",")
                res = self.lvalue_name()
                self.take("op", ")")
                self.emit("call", "PerformAT", 3)
                self.store(res)
                return
            n = self.args(")")
            self.emit("call", member, n + 1)
            if member in self.MUTATORS:
                if base_name is None:
                    raise Unsupported("mutator on a temporary")
                self.store(base_name)
                if not statement:
                    self.emit("load", self.vars[base_name])
                else:
                    return
            base_name = None


def cases_of(path):
    src = open(path).read()
    for m in re.finditer(r"^TEST\((\w+)\)\s*\{", src, flags=re.M):
        # matching brace
        i, depth = m.end(), 1
        while depth:
            depth += src[i] == "{"
            depth -= src[i] == "}"
            i += 1
        yield m.group(1), src.count("\n", 0, m.start()) + 1, src[m.end():i - 1]


def main():
    if not os.path.isdir(REF):
        sys.exit("needs the reference tree at " + REF)
    out = {"_comment": "generated by tests/golden/make_known_answer_vectors.py from the reference's unit tests "
                       "(FractalSharkTest/Test{HDRFloat,HDRFloatComplex,ATInfo,BLA}.cpp): operands, operation names of "
                       "tests/kat/kat_vm.hpp, expected values and tolerances",
           "cases": [], "skipped": []}
    for f in FILES:
        for name, line, body in cases_of(os.path.join(REF, f)):
            why = next((r for pat, r in SKIP_IF if pat in body), None)
            if why:
                out["skipped"].append({"name": name, "source": "%s:%d" % (f, line), "why": why})
                continue
            try:
                c = Compiler(tokenize(preprocess(body)))
                c.block(until=None)
                n_assert = sum(1 for op in c.ops if op[0].startswith("assert"))
                assert n_assert > 0
                out["cases"].append({"name": name, "source": "%s:%d" % (f, line), "slots": len(c.vars), "asserts": n_assert,
                                     "ops": c.ops})
            except (Unsupported, SyntaxError, KeyError, AssertionError) as e:
                out["skipped"].append({"name": name, "source": "%s:%d" % (f, line), "why": "not expressible: %r" % (e,)})
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=None, separators=(",", ":"))
        fh.write("\n")
    print("%d cases (%d assertions), %d skipped -> %s" % (len(out["cases"]), sum(c["asserts"] for c in out["cases"]),
                                                          len(out["skipped"]), OUT))
    for s in out["skipped"]:
        print("  skipped %-40s %s" % (s["name"], s["why"]))


if __name__ == "__main__":
    main()
