#!/usr/bin/env python
"""Per-layer HBM traffic of the conv launches of ONE full-size trunk pass of the bench process itself, from two
rocprofv3 --pmc runs of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pop512 --no-roofline`
(FETCH_SIZE and WRITE_SIZE in separate passes, each with --kernel-trace only, as MI355X_MICROARCH.md prescribes).
Units: the counters are KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of a wide streaming read at 64 B, so the
read side is doubled (guide, section HBM); WRITE_SIZE is taken as is (uncalibrated).  FETCH_SIZE counts the requests of
the XCDs' L2s to the fabric: Infinity-Cache hits are in it, so "traffic" is L2-miss traffic, an upper bound of HBM's.
A trunk pass = the conv dispatches in front of one k_head (back to the k_head before it); the LAST pass of the process is a
timed bench step at the full stream count.  A layer = its transform pass (template MODE 2 / 3 / 4 of k_conv_wino43), if it
has one, plus the convolution kernel that follows it; conv_block1 in one launch (k_conv_wino23r<.., FUSE1 = true>, with its
per-stream scale pass k_w23_mel_params) is ONE layer whose algorithmic bytes are the log-mel image in and the pooled map out.
Usage: python profiles/summarize_pmc_bench.py <fetch.db> <write.db> <n_streams> [out.json]"""
import json
import os
import re
import sqlite3
import sys

fetch_db, write_db, S = sys.argv[1], sys.argv[2], int(sys.argv[3])


def dispatches(path, counter):
    cur = sqlite3.connect(path).cursor()
    return cur.execute("select name, counter_value, duration from pmc_events where counter_name=? order by dispatch_id", (counter,)).fetchall()


def last_pass(rows):
    heads = [i for i, r in enumerate(rows) if "k_head" in r[0]]
    assert heads, "no trunk pass in the trace"
    lo = heads[-2] + 1 if len(heads) > 1 else 0
    return [r for r in rows[lo:heads[-1]] if "k_conv_wino" in r[0] or "k_conv3x3" in r[0] or "k_w23_mel_params" in r[0]]


def fused1(kname):
    m = re.search(r"k_conv_wino23r<([^>]*)>", kname)
    return bool(m) and m.group(1).split(",")[-1].strip() in ("true", "1")


def mode_of(kname):
    m = re.search(r"k_conv_wino43<([^>]*)>", kname)
    if not m:
        return 0
    args = [a.strip() for a in m.group(1).split(",")]
    return int(args[-1]) if len(args) >= 4 else 0   # <TTW, POOL, TRACE, MODE> (five arguments before round 4's clean-up: MODE last too)


def kind_of(kname):
    if "k_conv_wino23r" in kname:
        return "F(2x2,3x3), f16 hi + lo operands, weights resident in registers" + (", first conv computed into the patch ring" if fused1(kname) else "")
    if "k_conv_wino43s3" in kname:
        return "F(4x4,3x3), f16 hi + lo operands, six sweeps (128 x 128 tiles)"
    if "k_conv_wino43s2" in kname:
        return "F(4x4,3x3), f16 hi + lo operands, two sweeps (64 x 64 tiles)"
    if "k_conv_wino43s" in kname:
        return "F(4x4,3x3), f16 hi + lo operands"
    if "k_conv_wino43" in kname:
        return "F(4x4,3x3) f32" + (", hoisted transform" if mode_of(kname) == 1 else "")
    return "F(2x2,3x3) f32" if "wino" in kname else "direct f32"


f = last_pass(dispatches(fetch_db, "FETCH_SIZE"))
w = last_pass(dispatches(write_db, "WRITE_SIZE"))
assert len(f) == len(w) and [a[0] for a in f] == [b[0] for b in w], (len(f), len(w))
chans = [64, 128, 256, 512, 1024, 2048]
shapes = []
H, W = 469, 128
one_launch = any(fused1(r[0]) for r in f)
for b in range(6):
    for j in range(2):
        cin = (chans[b - 1] if b else 1) if j == 0 else chans[b]
        cout = chans[b]
        pool = j == 1 and b < 5
        if one_launch and b == 0:
            if j == 1:
                alg = 4.0 * (S * H * W * 1 + S * (H // 2) * (W // 2) * cout + 9 * 1 * 64 + 9 * 64 * cout)
                shapes.append((f"conv_block1 (one launch) {H}x{W} 1->64->{cout} pool", alg))
        elif cin % 8 == 0:
            Ho, Wo = (H // 2, W // 2) if pool else (H, W)
            alg = 4.0 * (S * H * W * cin + S * Ho * Wo * cout + 9 * cin * cout)
            shapes.append((f"conv_block{b + 1}.conv{j + 1} {H}x{W} {cin}->{cout}{' pool' if pool else ''}", alg))
    if b < 5:
        H, W = H // 2, W // 2
layers, cur = [], []
for i, (name, _, _) in enumerate(f):
    cur.append(i)
    if "k_w23_mel_params" in name:
        continue
    if mode_of(name) in (0, 1) or "k_conv_wino43s" in name or "wino43" not in name or "k_conv_wino23r" in name:
        # (round 6: a k_conv_wino43s layer is launched as several grids on several queues -- consecutive dispatches of the SAME
        # instantiation are one layer; two different layers never follow each other without a transform pass in between)
        if i + 1 < len(f) and f[i + 1][0] == name and "k_conv_wino43" in name:
            continue
        layers.append(cur)
        cur = []
assert not cur and len(layers) == len(shapes), (len(layers), len(shapes), [x[0][:40] for x in f])
print(f"{'layer':92s} {'fetch_GB(x2)':>12s} {'write_GB':>9s} {'traffic_GB':>10s} {'algorithmic_GB':>14s} {'ratio':>6s} {'ms':>7s}")
tf = tw = ta = 0.0
per_layer = []
for (name, alg), idx in zip(shapes, layers):
    fe = sum(f[j][1] for j in idx) * 1024 * 2 / 1e9
    wr = sum(w[j][1] for j in idx) * 1024 / 1e9
    ms = sum(f[j][2] for j in idx) / 1e6
    kern = kind_of(f[idx[-1]][0])
    label = f"{name} [{kern}]"
    print(f"{label:92s} {fe:12.3f} {wr:9.3f} {fe + wr:10.3f} {alg / 1e9:14.3f} {(fe + wr) / (alg / 1e9):6.2f} {ms:7.3f}")
    per_layer.append({"layer": name, "kernel": kern, "fetch_GB_x2": round(fe, 3), "write_GB": round(wr, 3), "algorithmic_GB": round(alg / 1e9, 3),
                      "ratio": round((fe + wr) / (alg / 1e9), 2), "ms_under_pmc": round(ms, 3)})
    tf += fe; tw += wr; ta += alg / 1e9
n = len(shapes)
print(f"{'total (' + str(n) + ' conv layers)':92s} {tf:12.3f} {tw:9.3f} {tf + tw:10.3f} {ta:14.3f} {(tf + tw) / ta:6.2f}")
print(f"per conv layer average: traffic {1e3 * (tf + tw) / n:.1f} MB, algorithmic {1e3 * ta / n:.1f} MB  (n_streams = {S})")
if len(sys.argv) > 4:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import kernel_source_hash
    json.dump({"_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (each with --kernel-trace only) on the bench "
                           "process itself (`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pop512 --no-roofline`): the conv "
                           "launches of its last trunk pass (a timed step, 512 streams).  Counters are KiB; FETCH_SIZE doubled per "
                           "MI355X_MICROARCH.md section HBM (gfx950 counts 128-B requests at 64 B) and includes Infinity-Cache hits (L2-miss "
                           "traffic); WRITE_SIZE as is.  bench.py quotes traffic_bytes_per_launch only while kernel_source_hash matches "
                           "the tree it runs from.",
               "kernel_source_hash": kernel_source_hash(), "n_streams": S, "launches": n, "fetch_GB_x2": tf, "write_GB": tw,
               "traffic_GB": tf + tw, "algorithmic_GB": ta, "traffic_bytes_per_launch": (tf + tw) * 1e9 / n,
               "algorithmic_bytes_per_launch": ta * 1e9 / n, "layers": per_layer},
              open(sys.argv[4], "w"), indent=1)
