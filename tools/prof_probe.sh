#!/bin/bash
# tools/prof_probe.sh <tag> <deep_probe args ...> — PMC passes for the standalone kernel harness tools/deep_probe
# (run on the GPU box).  Separate passes per counter group; the program itself follows `--`.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 "$@" --output-format csv -d $OUT/probe_${TAG}_$name -o p -- $REPO/tools/deep_probe "${ARGS[@]}" > $OUT/probe_${TAG}_$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/probe_${TAG}_$name.log; return 0; }
  echo "pass $name ok"
}
ARGS=("$@")
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run lds --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM
run mem --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_SMEM SQ_INST_CYCLES_VALU
python3 - "$TAG" <<'PY'
import collections, csv, glob, os, sys
tag = sys.argv[1]
G = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out")
for name in ("fetch", "write", "sq", "lds", "mem"):
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(G, "probe_%s_%s" % (tag, name), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "deep" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(vals.items()):
        print("%-8s %-26s mean %.6g over %d launches" % (name, c, sum(v) / len(v), len(v)))
PY
