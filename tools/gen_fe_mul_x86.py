"""Generates curdleproofs_pie_amd/csrc/fe_mul_x86.h: the host library's 6 x 64-bit Montgomery product (radix 2^384) as ONE block of
x86-64 inline assembly on mulx + adcx / adox (BMI2 + ADX): per row of the CIOS loop the low halves of the six partial products ride
the carry flag and the high halves the overflow flag, so the two carry chains of a row issue side by side and no product waits for
a flag.  The 7 accumulator words rotate through 7 registers (no moves between rows).  p < 2^381 keeps every row's top word inside
64 bits (the "no-carry" property the portable C version in host_g1.cpp relies on too).

    python tools/gen_fe_mul_x86.py
"""
import os

T = ["r8", "r9", "r10", "r11", "r12", "r13", "r14"]      # accumulator words, rotating
LO, HI = "rax", "rbx"


def row(lines, t, src):
    """t[0..6] += rdx * src[0..5]: low halves on the carry flag, high halves on the overflow flag"""
    lines.append("xorl %eax, %eax")                        # CF = OF = 0
    for j in range(6):
        lines.append("mulxq %d(%s), %%%s, %%%s" % (8 * j, src, LO, HI))
        lines.append("adcxq %%%s, %%%s" % (LO, t[j]))
        lines.append("adoxq %%%s, %%%s" % (HI, t[j + 1]))
    lines.append("adcq $0, %%%s" % t[6])                  # the low chain's last carry (the high chain cannot overflow: p < 2^381)


def sqr_text():
    """fe_sqr_adx: a^2 * 2^-384.  The 15 off-diagonal products a_i a_j (i < j) once (row by row, two finished words leaving for a
    12-word scratch after each row), doubled by a shift pass, the 6 squares a_i^2 added by one carry chain, then the same six
    reduction rows as the product and one addition of the high half: 57 mulx instead of 72."""
    L = []
    R = ["r8", "r9", "r10", "r11", "r12", "r13", "r14"]
    # ---- off-diagonal rows.  words t1..t10; live window in registers, finished pairs stored to {t}
    # row 0: a0 * a1..a5 -> t1..t6
    L.append("movq 0({a}), %rdx")
    L.append("mulxq 8({a}), %r8, %r9")            # t1, c
    L.append("mulxq 16({a}), %rax, %r10")
    L.append("addq %rax, %r9")                     # t2
    L.append("mulxq 24({a}), %rax, %r11")
    L.append("adcq %rax, %r10")                    # t3
    L.append("mulxq 32({a}), %rax, %r12")
    L.append("adcq %rax, %r11")                    # t4
    L.append("mulxq 40({a}), %rax, %r13")
    L.append("adcq %rax, %r12")                    # t5
    L.append("adcq $0, %r13")                      # t6
    L.append("movq %r8, 8({t})")
    L.append("movq %r9, 16({t})")
    # live: t3=r10 t4=r11 t5=r12 t6=r13.  row 1: a1 * a2..a5 -> t3..t7 (t7 = r8)
    L.append("movq 8({a}), %rdx")
    L.append("xorl %r8d, %r8d")                   # t7 = 0, CF = OF = 0
    for (off, lo_t, hi_t) in ((16, "r10", "r11"), (24, "r11", "r12"), (32, "r12", "r13"), (40, "r13", "r8")):
        L.append("mulxq %d({a}), %%rax, %%rbx" % off)
        L.append("adcxq %%rax, %%%s" % lo_t)
        L.append("adoxq %%rbx, %%%s" % hi_t)
    L.append("adcq $0, %r8")
    L.append("movq %r10, 24({t})")
    L.append("movq %r11, 32({t})")
    # live: t5=r12 t6=r13 t7=r8.  row 2: a2 * a3..a5 -> t5..t8 (t8 = r9)
    L.append("movq 16({a}), %rdx")
    L.append("xorl %r9d, %r9d")
    for (off, lo_t, hi_t) in ((24, "r12", "r13"), (32, "r13", "r8"), (40, "r8", "r9")):
        L.append("mulxq %d({a}), %%rax, %%rbx" % off)
        L.append("adcxq %%rax, %%%s" % lo_t)
        L.append("adoxq %%rbx, %%%s" % hi_t)
    L.append("adcq $0, %r9")
    L.append("movq %r12, 40({t})")
    L.append("movq %r13, 48({t})")
    # live: t7=r8 t8=r9.  row 3: a3 * a4, a5 -> t7..t9 (t9 = r10)
    L.append("movq 24({a}), %rdx")
    L.append("xorl %r10d, %r10d")
    for (off, lo_t, hi_t) in ((32, "r8", "r9"), (40, "r9", "r10")):
        L.append("mulxq %d({a}), %%rax, %%rbx" % off)
        L.append("adcxq %%rax, %%%s" % lo_t)
        L.append("adoxq %%rbx, %%%s" % hi_t)
    L.append("adcq $0, %r10")
    L.append("movq %r8, 56({t})")
    L.append("movq %r9, 64({t})")
    # live: t9=r10.  row 4: a4 * a5 -> t9, t10
    L.append("movq 32({a}), %rdx")
    L.append("mulxq 40({a}), %rax, %r11")
    L.append("addq %rax, %r10")
    L.append("adcq $0, %r11")
    L.append("movq %r10, 72({t})")
    L.append("movq %r11, 80({t})")
    # ---- double: t_k = (t_k << 1) | (t_{k-1} >> 63), k = 10 .. 1 (top down, so every word still sees its un-shifted lower neighbour);
    # t11 = t10 >> 63; t0 = 0.  (Adding the off-diagonal words twice on the adcx / adox chains instead was measured: slower, the two
    # chains serialise on each word.)
    L.append("movq 80({t}), %rax")
    L.append("shrq $63, %rax")
    L.append("movq %rax, 88({t})")
    for k in range(10, 1, -1):
        L.append("movq %d({t}), %%rax" % (8 * k))
        L.append("movq %d({t}), %%rbx" % (8 * (k - 1)))
        L.append("shldq $1, %rbx, %rax")
        L.append("movq %%rax, %d({t})" % (8 * k))
    L.append("shlq $1, 8({t})")
    L.append("movq $0, 0({t})")
    # ---- add the squares a_i^2 at words 2i, 2i+1: one carry chain (mulx and mov leave the flags alone)
    for i in range(6):
        L.append("movq %d({a}), %%rdx" % (8 * i))
        L.append("mulxq %rdx, %rax, %rbx")
        L.append(("addq" if i == 0 else "adcq") + " %%rax, %d({t})" % (16 * i))
        L.append("adcq %%rbx, %d({t})" % (16 * i + 8))
    # ---- Montgomery reduction of the low half: acc = T[0..5] (+ a top word), six rows acc = (acc + m p) / 2^64
    t = list(R)
    for j in range(6):
        L.append("movq %d({t}), %%%s" % (8 * j, t[j]))
    L.append("xorl %%%sd, %%%sd" % (t[6], t[6]))
    for i in range(6):
        L.append("movq %%%s, %%rdx" % t[0])
        L.append("imulq {pinv}, %rdx")
        row(L, t, "{p}")
        t = t[1:] + [t[0]]
    # ---- + the high half T[6..11]
    for j in range(6):
        L.append(("addq" if j == 0 else "adcq") + " %d({t}), %%%s" % (8 * (6 + j), t[j]))
    out_regs = t[:6]
    spare = [r for r in R if r not in out_regs]
    esc = lambda l: l.replace("%", "%%").replace("{a}", "%[a]").replace("{t}", "%[t]").replace("{p}", "%[p]").replace("{pinv}", "%[pinv]")
    body = "\n".join('      "%s\\n\\t"' % esc(l) for l in L)
    decl = "\n".join('  register uint64_t o%d __asm__("%s");' % (j, out_regs[j]) for j in range(6))
    outs = ", ".join('"=&r"(o%d)' % j for j in range(6))
    stores = " ".join("r[%d] = o%d;" % (j, j) for j in range(6))
    return '''
// Montgomery square a^2 * 2^-384 (57 mulx: 15 off-diagonal products, doubled, + 6 squares, + the six reduction rows).  Result < 2p.
static inline void fe_sqr_adx(uint64_t r[6], const uint64_t a[6], const uint64_t p[6], uint64_t pinv) {
  uint64_t T[12];
  uint64_t* t = T;
%s
  __asm__ __volatile__(
%s
      : %s
      : [a] "r"(a), [t] "r"(t), [p] "r"(p), [pinv] "m"(pinv)
      : "rax", "rbx", "rdx", "%s", "cc", "memory");
  %s
}
''' % (decl, body, outs, spare[0], stores), len(L)


def main():
    lines = []
    t = list(T)
    for r in T:
        lines.append("xorl %%%sd, %%%sd" % (r, r))
    for i in range(6):
        lines.append("movq %d({b}), %%rdx" % (8 * i))
        row(lines, t, "{a}")
        lines.append("movq %%%s, %%rdx" % t[0])             # m = t0 * (-p^-1) mod 2^64
        lines.append("imulq {pinv}, %rdx")
        row(lines, t, "{p}")                                # t += m * p: the low word vanishes
        t = t[1:] + [t[0]]                                  # ... and its register is the next row's (zero) top word
    out_regs = t[:6]
    spare = [r for r in T if r not in out_regs]
    esc = lambda l: l.replace("%", "%%").replace("{a}", "%[a]").replace("{b}", "%[b]").replace("{p}", "%[p]").replace("{pinv}", "%[pinv]")
    body = "\n".join('      "%s\\n\\t"' % esc(l) for l in lines)
    decl = "\n".join('  register uint64_t o%d __asm__("%s");' % (j, out_regs[j]) for j in range(6))
    outs = ", ".join('"=&r"(o%d)' % j for j in range(6))
    stores = " ".join("r[%d] = o%d;" % (j, j) for j in range(6))
    text = '''// GENERATED by tools/gen_fe_mul_x86.py -- do not edit.
// Montgomery product of the host library (6 x 64-bit limbs, radix 2^384) on mulx + adcx / adox.  Result < 2p (the caller subtracts p once).
// The six result words leave the block in the registers the rotation ends in (pinned register variables), so the block needs only
// three pointer registers besides its ten working registers: it still assembles with a frame pointer and under the sanitizers.
#pragma once
#include <cstdint>
#if defined(__x86_64__)
namespace cg1h {
static inline void fe_mul_adx(uint64_t r[6], const uint64_t a[6], const uint64_t b[6], const uint64_t p[6], uint64_t pinv) {
%s
  __asm__ __volatile__(
%s
      : %s
      : [a] "r"(a), [b] "r"(b), [p] "r"(p), [pinv] "m"(pinv)
      : "rax", "rbx", "rdx", "%s", "cc", "memory");
  %s
}
%s
}  // namespace cg1h
#endif
''' % (decl, body, outs, spare[0], stores, sqr_text()[0])
    path = os.path.join(os.path.dirname(__file__), "..", "curdleproofs_pie_amd", "csrc", "fe_mul_x86.h")
    with open(path, "w") as f:
        f.write(text)
    print("wrote", os.path.normpath(path), len(lines), "instructions")


if __name__ == "__main__":
    main()
