#!/usr/bin/env python
"""Generates st-ito_amd/csrc/conv_wino23r_body.inc: the group loop of k_conv_wino23r (conv_wino23r.hip) as an explicit
software pipeline -- one MFMA per slot, the other work dealt to the slots by hand, a scheduling fence after every slot.

Why generated: a workgroup is ONE wave per SIMD (512 registers each), so nothing but the wave's own instruction order hides
LDS latency, VALU dependency latency (8 cycles for a lone wave) or the matrix pipe: the first version of the kernel (transform,
then products, then epilogue, in program order) spent 19 000 cycles per pixel group against 3 072 of matrix pipe (profiles/
round4_w23_ablation.txt).  Here a group is 4 phases (one k-step of 16 input channels each) of 24 slots (4 positions x 2
channel halves x 3 products):

  slots  0..11  the B operands of positions 1, 2, 3 of THIS k-step from t (one (position, channel quad) unit per two slots:
                2 packed adds + 4 v_fma_mixlo/hi for the hi halves, then 4 for the lo halves) -- position j's registers are
                rewritten only after its products of the previous phase (slots 6 j .. 6 j + 5) and before its own (slot 6 j);
  slots 10..19  the next k-step's patch rows: 16 ds_read_b128 two columns ahead of the 16 packed fmas that turn them into t
                (t of the current k-step is dead after slot 11);
  slots 18..21  the B operands of position 0 of the next k-step;
  phase 1 / 3, slot 10: s_waitcnt vmcnt(0) + barrier (X), then the LDS-DMA copies of the k-steps whose ring entries that frees
                (phase 1: k-steps 2, 3 of the next group into this group's entries 0, 1; phase 3: k-steps 0, 1 of the group
                after next into this group's entries 2, 3): every copy has half a group to land, six entries suffice;
  phase 0, slots 0..17: the PREVIOUS group's outputs (its Z went to the exchange buffer behind its phase 3, with a barrier):
                reads, Y = sum_i A^T Z_i, BN + ReLU (+ pool), stores -- in the phase where the accumulators of positions 1..3
                are still dead, so that its temporaries cost no registers;
  after phase 3: the accumulators drain, Z_i of this group go to the exchange buffer, barrier.

    python tools/gen/gen_w23_body.py > st-ito_amd/csrc/conv_wino23r_body.inc"""
NB = 2
UNITS = [(1, 0), (1, 1), (2, 0), (2, 1), (3, 0), (3, 1)]   # (position j, channel quad) units of slots 0..11

def phase(ks):
    last = ks == 3
    sv = "sv"
    out = []
    for s in range(24):
        j, n, p = s // 6, (s % 6) // 3, s % 3
        items = []
        # ---- barrier + copies --------------------------------------------------------------------------------------
        if s == 10 and ks == 1:
            items.append("W23_X() if (more1) { W23_DMA_SETUP(gi + 1) W23_DMA_KSTEP(2, ent) W23_DMA_KSTEP(3, ent + 1) }")
        if s == 10 and ks == 3:
            items.append("W23_X() if (more2) { W23_DMA_SETUP(gi + 2) W23_DMA_KSTEP(0, ent + 2) W23_DMA_KSTEP(1, ent + 3) }")
            items.append("if (more1) { W23_NEXT_COORDS() }")
        # ---- B operands of positions 1..3 of this k-step -------------------------------------------------------------
        if s < 12:
            uj, uq = UNITS[s // 2]
            t = (s // 2) & 1
            if s % 2 == 0:
                items.append(f"W23_VV({uq}, {uj}, {t}) W23_SH({uq}, {uj}, {t}, {sv})")
            else:
                items.append(f"W23_SL({uq}, {uj}, {t}, {sv})")
        # ---- next k-step: loads two slots ahead of the row combinations ------------------------------------------------
        nxt = []
        order = [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 1), (1, 2), (1, 3)]
        if s == 10:
            nxt.append(f"W23_SETP(ent + {ks + 1})")
        if 10 <= s <= 17:
            q, b = order[s - 10]
            nxt.append(f"W23_LD({q}, {b})")
        if 12 <= s <= 19:
            q, b = order[s - 12]
            nxt.insert(0, f"W23_TT({q}, {b})")   # frees the load buffer the LD of this slot refills
        svn = "sv_n" if last else "sv"
        if s == 18: nxt.append(f"W23_VV(0, 0, 0) W23_SH(0, 0, 0, {svn})")
        if s == 19: nxt.append(f"W23_SL(0, 0, 0, {svn})")
        if s == 20: nxt.append(f"W23_VV(1, 0, 1) W23_SH(1, 0, 1, {svn})")
        if s == 21: nxt.append(f"W23_SL(1, 0, 1, {svn})")
        if nxt:
            items.append(("if (more1) { " + " ".join(nxt) + " }") if last else " ".join(nxt))
        # ---- the previous group's outputs ---------------------------------------------------------------------------
        # (phase 0: the accumulators of positions 1..3 are dead until their first product of the group -- slots 6, 12, 18 --
        # so the epilogue's temporaries cost no registers there)
        ep = {(0, 0): "W23_E_BEGIN() W23_E_SETUP(0)", (0, 1): "W23_E_LD(0)", (0, 3): "W23_E_Y(0)", (0, 4): "W23_E_LD(1)",
              (0, 6): "W23_E_Y(1)", (0, 8): "W23_E_ST(0)", (0, 9): "W23_E_SETUP(1)",
              (0, 10): "W23_E_LD(0)", (0, 12): "W23_E_Y(0)", (0, 13): "W23_E_LD(1)", (0, 15): "W23_E_Y(1)", (0, 17): "W23_E_ST(1)"}
        if (ks, s) in ep:
            items.append(ep[(ks, s)])   # unconditional (only the stores look at e_have): a temporary written under one branch and
                                        # read under another would be live around the whole loop for the register allocator
        out.append(f"        /* {ks}.{s:2d} */ W23_MF({j}, {n}, {p}, {ks}) " + " ".join(items) + " W23_FENCE()")
    return "\n".join(out)

print("// GENERATED by tools/gen/gen_w23_body.py -- do not edit (NB = 2, 4 k-steps per group)")
print("    for (int gi = g_lo; gi < g_hi; ++gi) {")
print("        const bool more1 = gi + 1 < g_hi, more2 = gi + 2 < g_hi;")
for ks in range(4):
    print(phase(ks))
print("        W23_ZSTORE()")
print("        W23_ROTATE()")
print("    }")
