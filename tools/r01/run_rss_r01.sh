#!/bin/bash
# peak resident memory of bin/gcn10 over a 10-block run (ru_maxrss of the child)
set -e
python tools/bench_pipeline.py --pattern natural --blocks 10 --modes files --keep --esa-compression 8 --workdir /tmp/gcn10_rss > gpurun_out/rss_plain.json
python3 - <<'PY'
import os, resource, subprocess
root = os.environ.get("GRAFT_REPO_ROOT", ".")
wd = "/tmp/gcn10_rss"
subprocess.run("rm -rf logs cn_rasters_drained cn_rasters_undrained", shell=True, cwd=wd)
p = subprocess.run([os.path.join(root, "bin", "gcn10"), "-c", "config.txt", "-o"], cwd=wd, capture_output=True)
print("rc", p.returncode, "peak RSS of the child: %.1f MB" % (resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1024.0))
PY
rm -rf /tmp/gcn10_rss
