"""Checks the one real batched-affine step of tools/maddbench29 (gpurun_out/affstep_lane0.txt: line 1 = x3, y3, pre, suf of lane 0; line 2 = the
XYZZ accumulator of madd29 for the same sum, all 256-bit Montgomery words): x3 ZZ == X and y3 ZZZ == Y (mod p)."""
import sys
p = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 1 << 256
Ri = pow(R, -1, p)
rows = [[int(w, 16) for w in l.split()] for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/affstep_lane0.txt")]
val = lambda ws, k: sum(ws[8 * k + i] << (32 * i) for i in range(8))
x3, y3 = val(rows[0], 0), val(rows[0], 1)
X, Y, ZZ, ZZZ = (val(rows[1], k) for k in range(4))
ok = (x3 * ZZ * Ri - X) % p == 0 and (y3 * ZZZ * Ri - Y) % p == 0
# and the point is 4 G of y^2 = x^3 + 3 (G = (1, 2))
x, y = x3 * Ri % p, y3 * Ri % p
on_curve = (y * y - x * x * x - 3) % p == 0
print("affine step == madd29 sum:", ok, " on curve:", on_curve)
sys.exit(0 if ok and on_curve else 1)
