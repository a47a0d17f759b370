import numpy as np, sys
dt = np.dtype([("temp","<f8"),("ref","<u4"),("chrpos","<u4"),("alignpos","<u4"),("glen","<u4"),("qs","<u4"),("pad","<u4")])
A = np.fromfile(sys.argv[1], dt); B = np.fromfile(sys.argv[2], dt)
print(len(A), len(B))
for f in dt.names:
    if f == "temp":
        d = A[f].view("u8") != B[f].view("u8")
    else:
        d = A[f] != B[f]
    print(f, int(d.sum()))
WA = np.fromfile(sys.argv[1]+".win", np.uint8).reshape(len(A), -1); WB = np.fromfile(sys.argv[2]+".win", np.uint8).reshape(len(B), -1)
bad = np.nonzero((A["temp"].view("u8") != B["temp"].view("u8")) | (A["alignpos"] != B["alignpos"]))[0]
print("bad", len(bad))
for h in bad[:12]:
    print(h, A[h], B[h], bytes(WA[h][:A[h]["glen"]]), bytes(WB[h][:B[h]["glen"]]))
