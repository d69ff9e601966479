# Round-5 experiment on the benchmark kernel's epilogue (VERDICT r04 "Next" #7): the two spline elements of a final-layer group
# evaluated together on packed arithmetic + binary bin descent (fused_common.hpp rqs_regs2) against the element-by-element
# evaluation of rounds 1-4 (lib/variants/epi_scalar.so = the same sources with -DNF_EPI_SCALAR).  Same process family, same box:
# the contract line's value / roofline, then one counter pass each (vector instructions, MFMA instructions, MFMA busy).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5epi; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
BP="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --no-graph --no-secondary"
cd $R
timeout 300 $B > $O/bench_pair.json 2> $O/bench_pair.err
NF_MI355X_LIB=$R/normalizing-flows_amd/lib/variants/epi_scalar.so timeout 300 $B > $O/bench_scalar.json 2> $O/bench_scalar.err
timeout 300 $B > $O/bench_pair2.json 2>> $O/bench_pair.err
cd /tmp && export TMPDIR=/tmp
C="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
timeout 170 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_pair -- $BP > $O/pmc_pair.log 2>&1; echo "pmc pair rc=$?"
NF_MI355X_LIB=$R/normalizing-flows_amd/lib/variants/epi_scalar.so timeout 170 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_scalar -- $BP > $O/pmc_scalar.log 2>&1; echo "pmc scalar rc=$?"
cd $R
python - <<'PY'
import csv, glob, json, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r5epi")
res = {}
for tag in ("pair", "scalar"):
    agg = collections.defaultdict(list)
    for path in glob.glob(os.path.join(O, "pmc_" + tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "rqs_fused_kernel<0, true" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    d = {"per_launch_mean": m}
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        d["mfma_busy"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024)
        # 2048 waves x 32 layer pairs per launch
        d["valu_instructions_per_layer_wave_incl_mfma"] = m.get("SQ_INSTS_VALU", 0) / (2048 * 32)
        d["mfma_instructions_per_layer_wave"] = m.get("SQ_INSTS_MFMA", 0) / (2048 * 32)
    try:
        b = json.loads([l for l in open(os.path.join(O, "bench_%s.json" % tag)) if l.startswith("{")][0])
        d["bench_value_rows_per_s"] = b["value"]; d["ms_per_step"] = b["ms_per_step"]; d["roofline_frac"] = b["roofline"]["frac"]
        d["chain_launch_ms"] = b["roofline"]["avg_launch_ms"]
    except Exception as e:
        d["bench_error"] = repr(e)
    res[tag] = d
try:
    b = json.loads([l for l in open(os.path.join(O, "bench_pair2.json")) if l.startswith("{")][0])
    res["pair_repeat"] = {"bench_value_rows_per_s": b["value"], "ms_per_step": b["ms_per_step"], "roofline_frac": b["roofline"]["frac"]}
except Exception as e:
    res["pair_repeat"] = repr(e)
json.dump(res, open(os.path.join(O, "epilogue_ab.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
