// kat_host.cpp -- the known-answer stack machine on the HOST build of the numeric headers (g++ -ffp-contract=off, the
// flags of host/libfsinputs.so).  C ABI for ctypes: kat_run_host(program, n, checks, max) -> assertions evaluated.
#include "kat_vm.hpp"

extern "C" int kat_run_host(const kat::Instr *prog, int n, kat::Check *out, int max_checks)
{
    static kat::Machine m; // (large: off the stack)
    m = kat::Machine{};
    return m.run(prog, n, out, max_checks);
}
