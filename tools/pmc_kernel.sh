#!/bin/bash
# SQ occupancy / stall counters of one osq:: kernel family inside a command (rocprofv3 --pmc, its own pass: no trace domains).
#   tools/pmc_kernel.sh <tag> <kernel-name-substring> <command...>
# WAIT_ANY (waves parked at s_waitcnt / barriers) + WAIT_INST_ANY (issue stalls) + ACTIVE_INST_ANY ~ WAVE_CYCLES
# (MI355X_MICROARCH.md, "rocprofv3 PMC slots").  Prints per-dispatch averages for the kernels whose name contains the substring.
tag=$1; pat=$2; shift 2
export TMPDIR=/tmp
out=/tmp/osq_pmc_$tag; rm -rf $out; mkdir -p $out
# PMC_COUNTERS="A B C ..." picks another set (at most 8 SQ counters per pass)
counters=${PMC_COUNTERS:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU}
# keep the counters this part has (an unknown name fails the whole pass); the list is cached for the box's lifetime
[ -s /tmp/osq_pmc_avail.txt ] || rocprofv3 -L > /tmp/osq_pmc_avail.txt 2>&1
have=""; for c in $counters; do if grep -qw "$c" /tmp/osq_pmc_avail.txt; then have="$have $c"; else echo "   (counter $c not available on this part)"; fi; done
counters=$have
rocprofv3 --pmc $counters -d $out -o r -- "$@" > $out/run.log 2>&1 || tail -5 $out/run.log
python - "$out" "$pat" <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection where kernel_name like ? "
                   "group by kernel_name, counter_name order by kernel_name, counter_name", ("%" + sys.argv[2] + "%",)).fetchall()
cur = None
vals = {}
for k, c, n, avg, tot in rows:
    k = k.split("(")[0][-60:]
    if k != cur:
        cur = k
        print("==", k, "dispatches", n)
        vals = {}
    vals[c] = avg
    print("   %-22s avg per dispatch %14.1f" % (c, avg))
    if c == "SQ_WAVE_CYCLES" and all(x in vals for x in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")):
        w = vals["SQ_WAVE_CYCLES"]
        print("   -> of wave cycles: parked (s_waitcnt / barrier) %.1f %%, issue-stalled %.1f %%, issuing %.1f %%; VALU share of issuing %.1f %%" % (
            100 * vals["SQ_WAIT_ANY"] / w, 100 * vals["SQ_WAIT_INST_ANY"] / w, 100 * vals["SQ_ACTIVE_INST_ANY"] / w,
            100 * vals.get("SQ_ACTIVE_INST_VALU", 0) / max(vals["SQ_ACTIVE_INST_ANY"], 1)))
PY
rm -rf $out
