#!/usr/bin/env python3
"""The fixed addition chain k_batch_decompress uses for y = a^((p+1)/4): a sliding window of width W = 4 over the constant
exponent, the eight odd powers a, a^3, ..., a^15 in LDS.  `chain()` returns (first_idx, ops) with ops = [(nsq, idx)]:
square nsq times, then multiply by a^(2 idx + 1) (idx = 0xFF: no multiplication).  tools/gen_consts.py writes it into
csrc/bls_consts.h (D_SQRT_CHAIN); bench.py takes the algorithmic v_mad_u64_u32 count of the kernel from `chain_cost()`.

    python tools/sqrt_chain.py        # compare window widths
"""
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
SQRT_EXP = (P + 1) // 4
WINDOW = 4


def chain(e=SQRT_EXP, w=WINDOW):
    ops, first, pending = [], None, 0
    i = e.bit_length() - 1
    while i >= 0:
        if not (e >> i) & 1:
            pending += 1
            i -= 1
            continue
        j = max(i - w + 1, 0)
        while not (e >> j) & 1:
            j += 1
        idx = (((e >> j) & ((1 << (i - j + 1)) - 1)) - 1) // 2
        if first is None:
            first = idx
        else:
            ops.append((pending + i - j + 1, idx))
        pending = 0
        i = j - 1
    if pending:
        ops.append((pending, 0xFF))
    return first, ops


def evaluate(a, e=SQRT_EXP, w=WINDOW, mod=P):
    """Run the chain on an integer (self-check: must equal pow(a, e, mod))."""
    first, ops = chain(e, w)
    tab = [pow(a, 2 * k + 1, mod) for k in range(1 << (w - 1))]
    r = tab[first]
    for nsq, idx in ops:
        for _ in range(nsq):
            r = r * r % mod
        if idx != 0xFF:
            r = r * tab[idx] % mod
    return r


def chain_cost(e=SQRT_EXP, w=WINDOW):
    """(squarings, products) including the table build (a^2, then 2^(w-1) - 1 products)."""
    first, ops = chain(e, w)
    return 1 + sum(n for n, _ in ops), (1 << (w - 1)) - 1 + sum(1 for _, i in ops if i != 0xFF)


if __name__ == "__main__":
    assert evaluate(0x1234567) == pow(0x1234567, SQRT_EXP, P)
    for w in (2, 3, 4, 5, 6):
        s, m = chain_cost(w=w)
        print(f"window {w}: {s} squarings + {m} products = {s * 301 + m * 392} v_mad_u64_u32")
