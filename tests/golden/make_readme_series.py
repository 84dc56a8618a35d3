"""Extracts the plotted closed-loop series of the reference's README example from its own result
figure (docs/src/assets/readme_result.svg, produced by `sim!(mpc, 40, [5, 0])`, README.md:66-74)
into tests/golden/readme_result_series.json.  Run in the build container (needs /root/reference);
the JSON is data: pixel coordinates of the five 40-sample polylines (y1, its set point, y2, its
upper bound, and the staircase of u)."""
import json, os, re, sys

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/docs/src/assets/readme_result.svg"
s = open(src).read()
series = []
for p in re.findall(r'<polyline[^>]*points="([^"]*)"', s):
    pts = [(float(a), float(b)) for a, b in re.findall(r'([-\d.]+)[ ,]([-\d.]+)', p)]
    if len(pts) >= 30:
        series.append(pts)
assert [len(p) for p in series] == [40, 40, 40, 40, 79], [len(p) for p in series]
out = {"source": "docs/src/assets/readme_result.svg (polylines with >= 30 points, document order)",
       "y1_px": [p[1] for p in series[0]], "ry1_px": series[1][0][1],
       "y2_px": [p[1] for p in series[2]], "y2max_px": series[3][0][1],
       "u_px": [series[4][0][1]] + [series[4][2 * k][1] for k in range(1, 40)],
       "anchors": "y1 = 0 at y1_px[0] (dead time), y1 = 5 at ry1_px; y2 = 0 at y2_px[0], y2 = 35 at y2max_px"}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "readme_result_series.json"), "w"), indent=0)
print("ok", len(out["u_px"]))
