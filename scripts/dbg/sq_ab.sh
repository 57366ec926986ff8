#!/bin/bash
# A/B of the SQ counters of the MAC kernels: Karatsuba (gc_mack_kernel) against the plain array (gc_mac_kernel)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sqab
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for k in 1 0; do
  export LGC_KARATSUBA=$k
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq1_$k -- python3 $R/scripts/gpu_probe.py big > $O/out1_$k.txt 2> $O/err1_$k.txt
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $O/sq2_$k -- python3 $R/scripts/gpu_probe.py big > $O/out2_$k.txt 2> $O/err2_$k.txt
done
python3 - <<PY
import csv, glob, collections
for k in (1, 0):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("sq1_%d" % k, "sq2_%d" % k):
        for f in glob.glob("$O/" + d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, cs in acc.items():
        if "mac" in kn:
            print("kara", k, kn, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(cs.items())}, "launches", len(list(cs.values())[0]))
PY
tail -3 $O/out1_1.txt $O/out1_0.txt
