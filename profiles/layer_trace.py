"""One prefill layer, dispatch by dispatch, from a rocprofv3 kernel trace (rocpd sqlite):
python profiles/layer_trace.py <results.db> [marker-kernel-substring]  -> the dispatches between two consecutive
launches of the marker kernel (default k_moe_sort = one per prefill layer), taken from the middle of the trace;
python profiles/layer_trace.py <db> <start-marker> <which> <end-marker>  -> from launch #which of the start marker to the
next launch of the end marker (e.g. k_vit_patchify 2 k_vit_pixel_shuffle = one whole ViT pass)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "k_moe_sort"
rows = c.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if marker in r[0]]
if len(idx) < 3:
    sys.exit(f"marker {marker!r} seen {len(idx)} times")
which = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) // 2
a, b = idx[which], idx[which + 1]
if len(sys.argv) > 4:                       # explicit end marker: first launch of it after the start marker
    b = next(i for i, r in enumerate(rows) if i > a and sys.argv[4] in r[0]) + 1
t0 = rows[a][1]
print(f"# dispatches {a}..{b} of {len(rows)} ({marker} #{which} to #{which + 1}); span {(rows[b][1] - t0) / 1e3:.1f} us")
print(f"{'start_us':>10} {'dur_us':>9} {'gap_us':>8}  kernel")
prev_end = None
for name, s, e in rows[a:b]:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.2f} {gap:8.2f}  {name[:120]}")
    prev_end = e
