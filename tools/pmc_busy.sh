#!/bin/bash
# where the wave cycles of each kernel go: tools/pmc_busy.sh [bytes] [single|wrapped] [decode|singlepass]
#   parked (s_waitcnt / barrier), issue-stalled, active; VALU / LDS share of the active ones; per wave
#   (SQ_* cycle counters count quad-cycles on gfx950: MI355X_MICROARCH.md)
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcb; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d /tmp/pmcb -o p -- python $R/tools/run_scan.py ${1:-1073741824} 3 ${2:-single} ${3:-} > /dev/null 2>&1
rm -rf /tmp/pmcb2; rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcb2 -o p -- python $R/tools/run_scan.py ${1:-1073741824} 3 ${2:-single} ${3:-} > /dev/null 2>&1
python - <<'PY'
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in ("/tmp/pmcb/p_counter_collection.csv", "/tmp/pmcb2/p_counter_collection.csv"):
    try:
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0]
            if "ffq::k_" in n and "synth" not in n:
                agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    except OSError:
        pass
for n, d in agg.items():
    m = {k: sum(v) / len(v) for k, v in d.items()}
    w = max(m.get("SQ_WAVES", 1), 1)
    wc = max(m.get("SQ_WAVE_CYCLES", 1), 1)
    print("%-44s waves %8.0f | wave cycles %9.0f per wave: parked %4.1f %% issue-stalled %4.1f %% active %4.1f %% (VALU %4.1f %% LDS %4.1f %% scalar %4.1f %% VMEM %4.1f %%; LDS-issue-stall %4.1f %%) | VALU insts/wave %6.0f SALU %6.0f LDS %5.0f | GUI_ACTIVE %10.0f SQ_BUSY %10.0f" % (
        n[:44], w, 4 * wc / w, 100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc,
        100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_LDS", 0) / wc,
        100 * m.get("SQ_ACTIVE_INST_SCA", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_VMEM", 0) / wc, 100 * m.get("SQ_WAIT_INST_LDS", 0) / wc,
        m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_SALU", 0) / w, m.get("SQ_INSTS_LDS", 0) / w, m.get("GRBM_GUI_ACTIVE", 0), m.get("SQ_BUSY_CYCLES", 0)))
PY
