#!/bin/bash
# tools/gpu.sh [--timeout S] -- '<command>' : gpurun, asked again while the pod has no free GPU slot (exit code 3 = nothing ran,
# nothing charged).  Never retries a command that ran.
for attempt in $(seq 1 40); do
    /usr/local/graft/bin/gpurun "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 45
done
exit 3
