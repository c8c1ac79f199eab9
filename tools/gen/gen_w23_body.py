#!/usr/bin/env python
"""Generates st-ito_amd/csrc/conv_wino23r_body.inc: the group loop of k_conv_wino23r (conv_wino23r.hip) as an explicit
software pipeline of inline-asm blocks, one matrix product per block.

Why generated, and why asm blocks.  A workgroup is ONE wave per SIMD (512 registers each), so nothing but the wave's own
instruction order hides LDS latency or the matrix pipe, and the wave issues ONE instruction per ~8 cycles whatever its type
(tools/ubench/w23_shadow.hip, profiles/round4_w23_shadow_ubench.txt: three plain VALU instructions hide under a 32-cycle
v_mfma_f32_32x32x16_f16, every further one costs 8 cycles; a packed-f32 instruction in the shadow of a product costs 12 - 20
cycles more; s_mov 8; ds_read_b128 16 from the third per product on; ds_write_b128 ~50 with four waves writing).  With ~900
instructions per pixel group against 96 products (3 072 cycles of matrix pipe) the loop is bound by instruction issue, so what
counts is the instruction count: hipcc puts an s_nop between any two inline-asm statements of which the second reads a
register the first defines (it has to assume a partial-register write), ~140 per group when every instruction is its own
statement -- here a block carries its product AND the 12 transform instructions of a (position, channel quad) unit, with the
one real hazard of that kind (v_fma_mixlo/hi_f16 writes half a register: one instruction between it and a reader) kept by the
order inside the block.

A group is 4 phases (one k-step of 16 input channels each) of 24 blocks (4 positions x [n0 n1] x 3 products, the two channel
halves alternating so that consecutive blocks never chain on one accumulator):

  blocks 0, 2, .. 10   the B operands of positions 1, 2, 3 of THIS k-step from t: unit (j, quad) = 4 adds (V = t[b1] +- t[b2]),
                       4 v_fma_mixlo/hi (hi halves), 4 more (lo halves).  Position j's registers are rewritten only after its
                       products of the previous phase (blocks 6 j .. 6 j + 5) and before its own (block 6 j);
  blocks 10 .. 17      the next k-step's patch rows: two ds_read_b128 per block (compiler-visible loads between the asm blocks),
  blocks 12 .. 19      ... and two blocks later the two packed fmas that turn them into t (t of this k-step is dead after block 10);
  blocks 20, 22        the B operands of position 0 of the next k-step;
  phase 1 / 3, block 10: s_waitcnt vmcnt(0) + barrier (X), then, one per block, the eight LDS-DMA copies whose ring entries that
                       frees (phase 1: k-steps 2, 3 of the next group into this group's entries 0, 1; phase 3: k-steps 0, 1 of the
                       group after next into this group's entries 2, 3): every copy has half a group to land, six entries suffice;
  phase 0, odd blocks 1 .. 17: the PREVIOUS group's outputs (its Z went to the exchange buffer behind its phase 3, with a barrier):
                       reads, Y = sum_i A^T Z_i, BN + ReLU (+ pool), stores -- in the phase where the accumulators of positions
                       1..3 are still dead, so that its temporaries cost no registers;
  after phase 3: the accumulators drain, Z_i of this group go to the exchange buffer, barrier.

    python tools/gen/gen_w23_body.py > st-ito_amd/csrc/conv_wino23r_body.inc"""
import sys

NB = 2
DMA_STEP = 1   # blocks between two copies (measured at 512 streams, conv_block1.conv2: 1 -> 5.40 ms, 4 -> 5.73 ms: the copies need the time to land, not a lower issue rate)
NO_T = NO_M = False   # ablation variants (W23_ABL & 1: no transform instructions, & 2: no products)
FUSE = False          # "fuse1" variant: the patch is not copied, it is COMPUTED -- relu(bn1(conv3x3(log-mel))) on the matrix pipe -- in the copy slots
VCOMB = {0: (0, 2, "sub"), 1: (1, 2, "add"), 2: (2, 1, "sub"), 3: (1, 3, "sub")}   # V(i, j) = t[a] -+ t[c]

class Asm:
    """One asm volatile statement: instruction strings with {name} operand references, operands registered by name."""
    def __init__(self):
        self.ins, self.outs, self.inps = [], [], []
    def out(self, name, cons, expr): self.outs.append((name, cons, expr)); return "%[" + name + "]"
    def inp(self, name, cons, expr): self.inps.append((name, cons, expr)); return "%[" + name + "]"
    def add(self, s): self.ins.append(s)
    def emit(self, indent="        "):
        if not self.ins: return indent + "/* (ablated) */"
        body = "\\n\\t".join(self.ins)
        o = ", ".join(f'[{n}] "{c}"({e})' for n, c, e in self.outs)
        i = ", ".join(f'[{n}] "{c}"({e})' for n, c, e in self.inps)
        return f'{indent}asm volatile("{body}"\n{indent}             : {o}\n{indent}             : {i});'

def mfma(a, j, n, p, ks):
    """product p of position j, channel half n: lo' hi, hi' lo, hi' hi (the large term last); the first product of a group
    starts the accumulator from the inline constant 0"""
    if NO_M: return
    w = f"Wt[{j}][{ks}][{n}][{1 if p == 0 else 0}]"
    b = f"W23_BL({j})" if p == 1 else f"W23_BH({j})"
    wa, ba = a.inp("w", "a", w), a.inp("b", "v", b)
    if ks == 0 and p == 0:
        acc = a.out("acc", "=&v", f"acc[{j}][{n}]")
        a.add(f"v_mfma_f32_32x32x16_f16 {acc}, {wa}, {ba}, 0")
    else:
        acc = a.out("acc", "+v", f"acc[{j}][{n}]")
        a.add(f"v_mfma_f32_32x32x16_f16 {acc}, {wa}, {ba}, {acc}")

def unit(a, q, j, sv):
    """B operand registers 2 q, 2 q + 1 of position j (hi and lo halves) from t[q]: 12 instructions"""
    if NO_T: return
    ta, tc, op = VCOMB[j]
    s = a.inp("sv", "s", sv)
    v = [a.out(f"v{e}", "=&v", f"vtmp[{e}]") for e in range(4)]
    for e in range(4):
        x = a.inp(f"ta{e}", "v", f"tt[{q}][{ta}][{e >> 1}][{e & 1}]")
        y = a.inp(f"tc{e}", "v", f"tt[{q}][{tc}][{e >> 1}][{e & 1}]")
        a.add(f"v_{op}_f32 {v[e]}, {x}, {y}")
    h = [a.out(f"h{r}", "=&v", f"bhv[{j}][{2 * q + r}]") for r in range(2)]
    l = [a.out(f"l{r}", "=&v", f"blv[{j}][{2 * q + r}]") for r in range(2)]
    a.add(f"v_fma_mixlo_f16 {h[0]}, {v[0]}, {s}, 0")
    a.add(f"v_fma_mixlo_f16 {h[1]}, {v[2]}, {s}, 0")
    a.add(f"v_fma_mixhi_f16 {h[0]}, {v[1]}, {s}, 0")
    a.add(f"v_fma_mixhi_f16 {h[1]}, {v[3]}, {s}, 0")
    a.add(f"v_fma_mixlo_f16 {l[0]}, {v[0]}, {s}, -{h[0]} op_sel:[0,0,0] op_sel_hi:[0,0,1]")
    a.add(f"v_fma_mixlo_f16 {l[1]}, {v[2]}, {s}, -{h[1]} op_sel:[0,0,0] op_sel_hi:[0,0,1]")
    a.add(f"v_fma_mixhi_f16 {l[0]}, {v[1]}, {s}, -{h[0]} op_sel:[0,0,1] op_sel_hi:[0,0,1]")
    a.add(f"v_fma_mixhi_f16 {l[1]}, {v[3]}, {s}, -{h[1]} op_sel:[0,0,1] op_sel_hi:[0,0,1]")

def tcomb(a, q, b):
    """t[q][b] = d[a1][b] + sg d[a2][b]: two packed fmas on the load buffer b & 1"""
    if NO_T: return
    sg = a.inp("sg", "v", "sg2")
    for h in range(2):
        t = a.out(f"t{h}", "=&v", f"tt[{q}][{b}][{h}]")   # early clobber: the second fma still reads its inputs
        d2 = a.inp(f"db{h}", "v", f"P2(dB[{b & 1}], {h})")
        d1 = a.inp(f"da{h}", "v", f"P2(dA[{b & 1}], {h})")
        a.add(f"v_pk_fma_f32 {t}, {d2}, {sg}, {d1}")

UNITS = [(1, 0), (1, 1), (2, 0), (2, 1), (3, 0), (3, 1)]   # (position j, channel quad) units of blocks 0, 2, .. 10
ORDER = [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 1), (1, 2), (1, 3)]   # (quad, patch column) of the loads / row combinations
EPI = {1: "W23_E_BEGIN() W23_E_SETUP(0)", 3: "W23_E_LD(0)", 5: "W23_E_Y(0) W23_E_LD(1)", 7: "W23_E_Y(1)", 9: "W23_E_ST(0) W23_E_SETUP(1)",
       11: "W23_E_LD(0)", 13: "W23_E_Y(0) W23_E_LD(1)", 15: "W23_E_Y(1)", 17: "W23_E_ST(1)"}

def phase(ks, out):
    last = ks == 3
    for s in range(24):
        j, p, n = s // 6, (s % 6) // 2, s % 2      # channel halves alternate
        pre, post = [], []
        a = Asm()
        mfma(a, j, n, p, ks)
        guard = None
        if s < 12 and s % 2 == 0:
            uj, uq = UNITS[s // 2]
            unit(a, uq, uj, "sv")
        if s == 0: pre.append(f"W23_STAMP({[0, 1, 4, 5][ks]})")
        if s == 10 and ks in (1, 3):
            pre.append(f"W23_STAMP({2 if ks == 1 else 6}) W23_X() W23_STAMP({3 if ks == 1 else 7})")
            if FUSE: pre.append("W23_C1_PREP(cB)" if ks == 1 else "W23_C1_PREP(cA)")
            else: pre.append("if (more1) { W23_DMA_PREP(cB) }" if ks == 1 else "if (more2) { W23_DMA_PREP(cA) }")
        # the eight copies behind a barrier, DMA_STEP blocks apart: 8 KB per wave issued back to back is more than a CU keeps in
        # flight (~20 KB): the issuing waves -- all of them -- stood still for ~1 400 cycles per burst
        for i in range(8):
            if FUSE: break
            at = 11 + DMA_STEP * i                      # block index counted from block 0 of the phase with the barrier
            if (ks, s) == ((1 + at // 24) % 4, at % 24):   # behind X of phase 1: k-steps 2, 3 of the next group
                post.append(f"if (more1) {{ W23_DMA_Q({2 + i // 4}, ent + {i // 4}, {i % 4}) }}")
            if (ks, s) == ((3 + at // 24) % 4, at % 24):   # behind X of phase 3: k-steps 0, 1 of the group after next
                if at < 24: post.append(f"if (more2) {{ W23_DMA_Q({i // 4}, ent + {2 + i // 4}, {i % 4}) }}")
                else:       # ... continued in the next trip of the loop: that group is now "the next one", ent has moved on by 4
                    post.append(f"if (e_have && more1) {{ W23_DMA_Q({i // 4}, ent + {4 + i // 4}, {i % 4}) }}")
        # fuse1: this wave's parity plane (51 pixels = two column blocks of 32) x 32 first-conv channels (phase 1: channels 32..63 =
        # k-steps 2, 3 of the next group into this group's entries 0, 1; phase 3: channels 0..31 = k-steps 0, 1 of the group after
        # next into entries 2, 3) as 2 x 3 products [32 channels x 16 window taps] x [16 taps x 32 pixels]; an accumulator is read
        # (ReLU, 4 ds_write_b128) two blocks behind its products: two products of the main stream have issued in between, each
        # waits for the matrix pipe, so the first-conv products have left it.  Behind the phase-1 barrier also the log-mel window of
        # the group after next (LDS-DMA, <= 3 rows per wave).
        if FUSE and ks in (1, 3):
            mb, e0, tg = (1, 0, "mel_pB") if ks == 1 else (0, 2, "1 - mel_pB")
            items = {10: f"W23_C1_A({mb}) W23_C1_MLD(0, {tg})", 11: "W23_C1_B(0)", 12: "W23_C1_MM(0)", 13: f"W23_C1_MM(1) W23_C1_MLD(1, {tg})",
                     14: "W23_C1_MM(2)", 15: "W23_C1_B(1)", 16: f"W23_C1_ST(0, ent + {e0}) W23_C1_MM(0)", 17: "W23_C1_MM(1)", 18: "W23_C1_MM(2)",
                     21: f"W23_C1_ST(1, ent + {e0})"}
            if s in items:   # unconditional like the look-ahead: behind the last groups it runs on stale windows into entries nobody reads
                if ks == 1 and s == 10: post.append("if (more2) { W23_MEL_DMA(cA, 1 - mel_pB) }")
                post.append(items[s])
        nxt_pre = []
        if s == 10: nxt_pre.append(f"W23_SETP(ent + {ks + 1})")
        if 10 <= s <= 17:
            q, b = ORDER[s - 10]
            nxt_post = None if NO_T else f"W23_LD({q}, {b})"
        else:
            nxt_post = None
        if 12 <= s <= 19:
            q, b = ORDER[s - 12]
            tcomb(a, q, b)
        if s in (20, 22):
            unit(a, (s - 20) // 2, 0, "sv_n" if last else "sv")
        if ks == 0 and s in EPI: post.append(EPI[s])
        if last and s == 11: post.append("if (more1) { W23_NEXT_SV() }")
        lines = list(pre)
        # (the last phase works ahead for the NEXT group; behind the last group that work runs on stale LDS contents and is never
        # used -- cheaper than a branch around it, which would also make hipcc copy the accumulators between the two arms)
        if nxt_pre: lines.append(" ".join(nxt_pre))
        if guard:   # the last phase works ahead for the NEXT group: without one, only the product
            b_ = Asm(); mfma(b_, j, n, p, ks)
            lines.append(f"if ({guard}) {{\n" + a.emit("            ") + "\n        } else {\n" + b_.emit("            ") + "\n        }")
        else:
            lines.append(a.emit().lstrip())
        if nxt_post: lines.append(nxt_post)
        lines += post
        out.append(f"        /* {ks}.{s:2d} */ " + "\n        ".join(lines) + "\n        W23_FENCE()")

def body(proto):
    out = []
    if proto:
        # t and the position-0 operands of the first group's first k-step (no products yet)
        for q, b in ORDER:
            if not NO_T: out.append(f"    W23_LD({q}, {b})")
            a = Asm(); tcomb(a, q, b); out.append(a.emit("    "))
        for q in range(2):
            a = Asm(); unit(a, q, 0, "sv"); out.append(a.emit("    "))
        return out
    out.append("    for (int gi = g_lo; gi < g_hi; ++gi) {")
    out.append("        const bool more1 = gi + 1 < g_hi, more2 = gi + 2 < g_hi;")
    for ks in range(4):
        phase(ks, out)
    out.append("        W23_ZSTORE()")
    out.append("        W23_ROTATE()")
    out.append("    }")
    return out

proto = "prologue" in sys.argv[1:]
FUSE = "fuse1" in sys.argv[1:]
print("// GENERATED by tools/gen/gen_w23_body.py -- do not edit (NB = 2, 4 k-steps per group; the W23_ABL & 3 variants are timing experiments)")
for abl in range(4):
    NO_T, NO_M = bool(abl & 1), bool(abl & 2)
    print(("#if" if abl == 0 else "#elif") + f" (W23_ABL & 3) == {abl}")
    print("\n".join(body(proto)))
print("#endif")
