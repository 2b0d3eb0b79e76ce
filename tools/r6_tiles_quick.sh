cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6t1
timeout 600 python tools/sell_probe.py --scale 22 --pr-scale 22 --variants tiles --oracle > gpurun_out/r6t1/probe_s22.jsonl 2> gpurun_out/r6t1/probe_s22.err; echo rc=$?
cut -c1-330 gpurun_out/r6t1/probe_s22.jsonl; tail -3 gpurun_out/r6t1/probe_s22.err
