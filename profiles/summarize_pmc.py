#!/usr/bin/env python
"""Per-dispatch HBM traffic of the conv kernels from two rocprofv3 --pmc runs (FETCH_SIZE and
WRITE_SIZE collected in separate passes, as MI355X_MICROARCH.md prescribes).  Units: the counters
are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of a wide streaming read at 64 B, so
the read side is doubled (guide section HBM); WRITE_SIZE is taken as is (uncalibrated).
Usage: python profiles/summarize_pmc.py <fetch.db> <write.db> <n_streams>"""
import sqlite3
import sys

fetch_db, write_db, S = sys.argv[1], sys.argv[2], int(sys.argv[3])


def per_dispatch(path, counter):
    cur = sqlite3.connect(path).cursor()
    return cur.execute("select name, counter_value, duration from pmc_events where counter_name=? and (name like '%k_conv3x3%' or name like '%k_conv_wino%') "
                       "order by dispatch_id", (counter,)).fetchall()


f = per_dispatch(fetch_db, "FETCH_SIZE")
w = per_dispatch(write_db, "WRITE_SIZE")
assert len(f) == len(w), (len(f), len(w))
chans = [64, 128, 256, 512, 1024, 2048]
layers = []
H, W = 469, 128
for b in range(6):
    for j in range(2):
        cin = (chans[b - 1] if b else 1) if j == 0 else chans[b]
        cout = chans[b]
        pool = j == 1 and b < 5
        if cin % 8 == 0:
            Ho, Wo = (H // 2, W // 2) if pool else (H, W)
            alg = 4.0 * (S * H * W * cin + S * Ho * Wo * cout + 9 * cin * cout)
            layers.append((f"{H}x{W} {cin}->{cout}{' pool' if pool else ''}", alg, 2.0 * 9 * cin * cout * H * W * S))
    if b < 5:
        H, W = H // 2, W // 2
print(f"{'layer':46s} {'fetch_GB(x2)':>12s} {'write_GB':>9s} {'traffic_GB':>10s} {'algorithmic_GB':>14s} {'ratio':>6s} {'FLOP/B':>7s}")
tf = tw = ta = 0.0
# conv_bench runs every layer twice (warm + 1 rep); a layer is one dispatch, or two when the F(4x4,3x3) input transform is
# hoisted (template MODE 2 = transform pass, then MODE 1 = convolution): keep the dispatches of the second call of each layer
def mode_of(kname):
    if "wino43" not in kname:
        return 0
    args = kname[kname.index("<") + 1:kname.index(">")].split(",")
    return int(args[4]) if len(args) > 4 else 0
pos = 0
for i, (name, alg, fl) in enumerate(layers):
    per_call = 2 if mode_of(f[pos][0]) == 2 else 1
    idx = range(pos + per_call, pos + 2 * per_call)
    pos += 2 * per_call
    fe = sum(f[j][1] for j in idx) * 1024 * 2 / 1e9
    wr = sum(w[j][1] for j in idx) * 1024 / 1e9
    kn = f[idx[-1]][0]
    kern = ("F(4x4,3x3) + hoisted input transform" if per_call == 2 else "F(4x4,3x3)") if "wino43" in kn else ("F(2x2,3x3)" if "wino" in kn else "direct")
    name = f"{name} [{kern}]"
    print(f"{name:46s} {fe:12.3f} {wr:9.3f} {fe + wr:10.3f} {alg / 1e9:14.3f} {(fe + wr) / (alg / 1e9):6.2f} {fl / ((fe + wr) * 1e9):7.0f}")
    tf += fe; tw += wr; ta += alg / 1e9
assert pos == len(f), (pos, len(f))
print(f"{'total (11 conv layers)':46s} {tf:12.3f} {tw:9.3f} {tf + tw:10.3f} {ta:14.3f} {(tf + tw) / ta:6.2f}")
print(f"per conv layer average: traffic {1e3 * (tf + tw) / 11:.1f} MB, algorithmic {1e3 * ta / 11:.1f} MB  (n_streams = {S})")
if len(sys.argv) > 4:
    import json
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import kernel_source_hash
    json.dump({"_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (each with --kernel-trace only) on "
                           "`python tools/conv_bench.py --streams 512 --reps 1 --modes 9` (the production mix of algorithms); counters are KiB; FETCH_SIZE doubled per "
                           "MI355X_MICROARCH.md section HBM (gfx950 counts 128-B requests at 64 B); WRITE_SIZE as is.  bench.py quotes "
                           "traffic_bytes_per_launch only while kernel_source_hash matches the tree it runs from.",
               "kernel_source_hash": kernel_source_hash(), "n_streams": S, "launches": 11, "fetch_GB_x2": tf, "write_GB": tw, "traffic_GB": tf + tw, "algorithmic_GB": ta,
               "traffic_bytes_per_launch": (tf + tw) * 1e9 / 11, "algorithmic_bytes_per_launch": ta * 1e9 / 11},
              open(sys.argv[4], "w"), indent=1)
