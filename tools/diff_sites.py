import json,sys
A=[json.loads(l) for l in open(sys.argv[1])]; B=[json.loads(l) for l in open(sys.argv[2])]
key=lambda s:(s["primer"],s["on_for"],s["ref"],s["pos"])
da={key(s):s for s in A}; db={key(s):s for s in B}
print(len(A),len(B),len(da),len(db))
onlya=[k for k in da if k not in db]; onlyb=[k for k in db if k not in da]
print("only wave",len(onlya),"only seq",len(onlyb))
for k in onlya[:6]: print("W",da[k])
for k in onlyb[:6]: print("S",db[k])
dt=[k for k in da if k in db and da[k]!=db[k]]
print("differing",len(dt))
for k in dt[:5]: print(da[k],db[k])
