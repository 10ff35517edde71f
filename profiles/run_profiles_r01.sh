# Round-1 profile recipe (run on the GPU box through gpurun): kernel trace + two PMC passes of
# the default bench command, all with --no-cpu-baseline (no child processes under rocprofv3).
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_kt.log 2>&1
echo kt done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_write.log 2>&1
echo write done

# summaries -> profiles/ (the files the bench line and DESIGN.md cite)
python3 - <<'PY'
import collections, csv, glob, json, os, re
R = os.environ["GRAFT_REPO_ROOT"]
P = os.path.join(R, "profiles")
ks = max(glob.glob(R + "/gpurun_out/prof_kt/*/*kernel_stats.csv"), key=os.path.getmtime)
open(os.path.join(P, "r01_kernel_stats_bench_config2.csv"), "w").write(open(ks).read())
avg = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(ks))}
sums = {}
for counter, d in (("FETCH_SIZE", "prof_fetch"), ("WRITE_SIZE", "prof_write")):
    f = max(glob.glob(R + "/gpurun_out/%s/*/*counter_collection.csv" % d), key=os.path.getmtime)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    with open(os.path.join(P, "r01_pmc_%s_bench_config2.csv" % counter.split("_")[0].lower()), "w") as out:
        out.write("Kernel_Name,Counter_Name,Dispatches,Mean_Counter_Value_KiB,Min,Max\n")
        for k, v in per.items():
            vals = list(v.values())
            out.write('"%s",%s,%d,%.3f,%.3f,%.3f\n' % (k, counter, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
            sums.setdefault(k, {})[counter] = sum(vals) / len(vals)
traffic = {}
for name, key in (("cn_strip_kernel<1, 1", "config2"), ("cn_strip_kernel<0, 3", "config4")):
    for k, v in sums.items():
        if name in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            short = re.search(r"cn_strip_kernel<[^>]*>", k).group(0)
            traffic[key] = {"kernel": short, "FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                            "hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024),
                            "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads -> doubled; WRITE_SIZE exact "
                                          "for 16-B/lane stores; both in KiB (MI355X_MICROARCH.md, HBM section)",
                            "rocprof_avg_ns": next((a for n, a in avg.items() if name in n), None)}
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(traffic)[:400])
PY
python3 $R/bench.py > $R/gpurun_out/bench_final.json
cp $R/gpurun_out/bench_final.json $R/gpurun_out/r01_bench_config2.json
cat $R/gpurun_out/bench_final.json | cut -c1-300
