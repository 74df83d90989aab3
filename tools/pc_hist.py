"""histogram of a rocprofv3 PC-sampling csv by (code object id, offset); prints the raw columns first"""
import csv, sys, collections
f = sys.argv[1]
rd = csv.DictReader(open(f))
print("columns:", rd.fieldnames)
h = collections.Counter()
n = 0
for r in rd:
    n += 1
    key = (r.get("Code_Object_Id") or r.get("code_object_id"), r.get("Code_Object_Offset") or r.get("code_object_offset"), r.get("Instruction") or "")
    h[key] += 1
print("samples", n)
for (co, off, ins), c in sorted(h.items(), key=lambda kv: -kv[1])[:400]:
    print(co, off, c, ins)
