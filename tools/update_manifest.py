"""profiles/MANIFEST.json: record the commit the round's profile files were taken at.  usage: update_manifest.py r05 <commit> [made_by per file defaults]"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, commit = sys.argv[1], sys.argv[2]
p = os.path.join(root, "profiles", "MANIFEST.json")
m = json.load(open(p))
for f in sorted(os.listdir(os.path.join(root, "profiles"))):
    if f.startswith(tag + "_"):
        made = "python bench.py" if f == f"{tag}_bench_line.json" else (m.get(f, {}).get("made_by") or f"tools/make_profiles.sh {tag}")
        m[f] = {"commit": commit, "made_by": made}
json.dump(m, open(p, "w"), indent=1)
print(len([k for k in m if k.startswith(tag + "_")]), "entries for", tag)
