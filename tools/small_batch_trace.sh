# GPU-side timeline of one blocking call on a small batch (kernel durations and the gaps between them).
# bash tools/small_batch_trace.sh [bytes]
set -eo pipefail
export TMPDIR=/tmp
OUT=gpurun_out/small_trace
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 120 rocprofv3 --kernel-trace -f csv -d $OUT -o t -- python3 bench.py --bytes ${1:-1000000} --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/log.txt
python3 - $OUT <<'PY'
import csv, re, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [(re.search(r"(k_\w+)", r["Kernel_Name"]) or [None, r["Kernel_Name"][:16]])[1] for r in rows]
first = "k_small_batch" if "k_small_batch" in names else "k_rec_lo"
idx = [i for i, n in enumerate(names) if n == first]
for s, e in zip(idx[40:43], idx[41:44]):
    prev = None; line = []
    for i in range(s, e):
        st, en = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        line.append("%s %.1f(+%.1f)" % (names[i][2:10], (en - st) / 1e3, ((st - prev) / 1e3) if prev else 0)); prev = en
    print(" ".join(line))
    print("span %.1f us, call to call %.1f us" % ((prev - int(rows[s]["Start_Timestamp"])) / 1e3,
                                                  (int(rows[e]["Start_Timestamp"]) - int(rows[s]["Start_Timestamp"])) / 1e3))
PY
