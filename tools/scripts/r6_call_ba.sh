#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/seq_prof
rm -rf $O
timeout 250 rocprofv3 --kernel-trace --output-format csv -d $O/maf -- python $GRAFT_REPO_ROOT/tools/maf_density_profile.py --steps 1 > /dev/null 2>&1; echo "rc=$?"
timeout 250 rocprofv3 --kernel-trace --output-format csv -d $O/glow -- python $GRAFT_REPO_ROOT/tools/glow_copy_audit.py > /dev/null 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for tag in ("maf", "glow"):
    f = glob.glob("gpurun_out/seq_prof/%s/**/*kernel_trace.csv" % tag, recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    names = [(r["Kernel_Name"][:48], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
              r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")) for r in rows]
    n = len(names)
    seg = names[-(260 if tag == "maf" else 420):]
    open("gpurun_out/seq_%s.txt" % tag, "w").write("\n".join("%-50s %8.1f us grid %s wg %s" % x for x in seg))
PY
rm -rf $O
