# CLI tests (incl. the lookup subset), then the full 2651-block queue of the reference's shapefile, one box.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_cli.py tests/test_gpu_parity.py::test_tune_single_raster_picks_a_position_and_changes_no_result -x -q > gpurun_out/cli_r02.log 2>&1 || { tail -40 gpurun_out/cli_r02.log; exit 1; }
tail -3 gpurun_out/cli_r02.log
timeout -k 10 2400 python3 tools/run_full_queue.py > gpurun_out/full_queue_r02.json 2> gpurun_out/full_queue_r02.err || { tail -5 gpurun_out/full_queue_r02.err; cut -c1-3000 gpurun_out/full_queue_r02.json; exit 1; }
cut -c1-2500 gpurun_out/full_queue_r02.json
