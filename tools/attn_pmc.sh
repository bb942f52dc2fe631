#!/bin/bash
# rocprofv3 --pmc passes (own runs, --kernel-trace only) on the attention forward harness: bash tools/attn_pmc.sh <tag> <libname> [libname ...]
# (library = tools/ubench/_var/libattn_<name>.so, timed against itself so that every dispatch of a pass is the same kernel)
cd "$(dirname "$0")/.."
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/attn_pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
for lib in "$@"; do
  L=$PWD/tools/ubench/_var/libattn_$lib.so
  for pass in "a:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" "b:GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "c:FETCH_SIZE" "d:WRITE_SIZE"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    RAW=/tmp/attn_pmc_raw/$lib/$name; rm -rf $RAW; mkdir -p $RAW
    ATTN_OLD=$L ATTN_NEW=$L rocprofv3 --kernel-trace --pmc $ctrs -d $RAW -o a --output-format csv -- tools/ubench/attn_ab.bin time > $OUT/${lib}_$name.log 2>&1
    cp $(find $RAW -name "*counter_collection.csv" | head -1) $OUT/${lib}_$name.csv 2>/dev/null
  done
done
python3 - "$OUT" "$@" <<'PY'
import collections, csv, glob, json, os, sys
out, libs = sys.argv[1], sys.argv[2:]
res = {}
for lib in libs:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f"{out}/{lib}_?.csv")):
        for r in csv.DictReader(open(f)):
            if "attn_bidir_fwd" not in r["Kernel_Name"]:
                continue
            tmpl = r["Kernel_Name"].split("<")[1].split(">")[0]
            key = f"<{tmpl}> grid {r['Grid_Size']}"
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
                agg[key]["_dur_ns_" + r["Counter_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    res[lib] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
for lib, ks in res.items():
    for k, c in ks.items():
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        print(lib, k, "dur_us %.0f" % (c.get("_dur_ns_SQ_WAVE_CYCLES", 0) / 1e3), "wait_any %.3f wait_inst %.3f active %.3f wait_lds %.3f mfma_busy/wavecyc*4 %.3f valu_insts %.3g lds_conf %.3g fetch %.3g write %.3g" % (
            c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_WAIT_INST_LDS", 0) / wc,
            c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * wc), c.get("SQ_INSTS_VALU", 0), c.get("SQ_LDS_BANK_CONFLICT", 0), c.get("FETCH_SIZE", 0), c.get("WRITE_SIZE", 0)))
PY
