#!/usr/bin/env python3
"""Cost of y = a^((p+1)/4) as fp28.h's fp_pow6 evaluates it (sliding window of width W over the constant exponent,
odd powers a, a^3, ..., a^(2^W - 1) in registers): (squarings, products).  Used by bench.py to state the algorithmic
v_mad_u64_u32 count of k_batch_decompress, and to compare window widths.

    python tools/sqrt_chain.py
"""
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
SQRT_EXP = (P + 1) // 4
WINDOW = 3          # must match fp_pow6 in curdleproofs_pie_amd/csrc/fp28.h


def chain_cost(e=SQRT_EXP, w=WINDOW):
    """Mirror of fp_pow6: returns (squarings, products) including the table build."""
    nsqr, nmul = 1, (1 << (w - 1)) - 1          # a^2, then a^3, a^5, ... by repeated products with a^2
    i = e.bit_length() - 1
    started = False
    while i >= 0:
        if not (e >> i) & 1:
            nsqr += 1
            i -= 1
            continue
        j = max(i - w + 1, 0)
        while not (e >> j) & 1:
            j += 1
        if started:
            nsqr += i - j + 1
            nmul += 1
        started = True
        i = j - 1
    return nsqr, nmul


if __name__ == "__main__":
    for w in (2, 3, 4, 5, 6):
        s, m = chain_cost(w=w)
        print(f"window {w}: {s} squarings + {m} products = {s * 301 + m * 392} v_mad_u64_u32")
