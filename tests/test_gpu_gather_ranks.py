"""The N > 1 branch of the C-ABI gather (include/havac_dev.h level 3, havac_amd/csrc/havac_gather.hip), executed.

RCCL refuses two ranks on one device and this pool hands out one-GPU boxes, so until round 5 the grouped ncclSend / ncclRecv
of havac_gather_records had never run anywhere.  Here 2 and 4 processes share GPU 0 and go through
the REAL havac_gather_counts / havac_gather_records / havac_gather_wait, bound to tests/native/librccl_standin.so (the eleven
RCCL entry points over shared memory) by havac_gather_use_library -- an argument, not an environment variable.  The gathered
list must equal the CPU checker's (the emission order of device/HavacHls.cpp:151-152,264; one device per object in the
reference, host/Havac.hpp:51).  tests/gather_rank_worker.py is one rank.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

STANDIN_SRC = os.path.join(ROOT, "tests", "native", "rccl_standin.cpp")
STANDIN = os.path.join(ROOT, "tests", "native", "librccl_standin.so")


def build_standin():
    if not os.path.isfile(STANDIN) or os.path.getmtime(STANDIN) < os.path.getmtime(STANDIN_SRC):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", STANDIN, STANDIN_SRC, "-lrt"], check=True)
    return STANDIN


@pytest.mark.parametrize("world", [2, 4])
def test_c_abi_gather_between_ranks_on_one_gpu(tmp_path, oracle, world):
    build_standin()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    worker = os.path.join(ROOT, "tests", "gather_rank_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(tmp_path)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=600)[0])
    finally:
        for p in procs:          # the exact processes this test started, nothing else
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r}:\n{logs[r][-3000:]}"
    res = [json.load(open(tmp_path / f"result_{r}.json")) for r in range(world)]
    # 1. the sharded pass: rank 0's list is the checker's, element for element; every rank reported only its own columns
    first = res[0]["pass"]
    assert first["equals_oracle"] and first["tail_untouched"] and first["records"] > 300
    assert first["counts"] == [r["pass"]["found"] for r in res] and sum(first["counts"]) == first["records"]
    assert all(r["pass"]["inside_my_columns"] and r["pass"]["counts"] == first["counts"] for r in res)
    assert min(first["counts"]) > 0
    # 2. ragged lists: offsets, zero-length ranks, rank 0 empty, many chunks
    for r in res:
        assert r["lists"] == {"mixed": True, "all_in_last": True, "rank0_empty": True, "nothing": True, "big": True}, r["lists"]
    # 3. the -1 sentinel: every rank sees it, every rank refuses the records with the failed rank's number, and the next gather works
    for r in res:
        assert r["failed_rank"]["counts"] == [5] * (world - 1) + [-1]
        assert f"rank {world - 1}" in r["failed_rank"]["message"]
        assert r["after_failed_rank"] == [2] * world
    # 4. a receive buffer that is too small: HAVAC_E_LENGTH on rank 0 before anything is posted; the senders' operations cannot
    #    complete (RCCL: for ever; the stand-in: until its 1.5 s deadline); every communicator is unusable afterwards
    assert res[0]["refused"]["kind"] == "length" and "receive buffer of 10 records" in res[0]["refused"]["message"]
    for r in res[1:]:
        assert r["refused"]["kind"] == "runtime" and "ncclGroupEnd" in r["refused"]["message"] and 1.0 < r["refused"]["seconds"] < 30
    assert all(r["after_refused"] == "LogicError" for r in res)
    # 5. the deadline: an exchange still in flight after 300 ms -> HAVAC_E_TIMEOUT with the rank and the stage in the message
    for k, r in enumerate(res):
        assert r["deadline_not_reached"] == list(range(world))
        assert f"rank {k} of {world}" in r["deadline"]["message"] and "ncclAllGather" in r["deadline"]["message"]
        assert "300 ms" in r["deadline"]["message"] and 0.25 < r["deadline"]["seconds"] < 1.4
    assert all(r["version"] == 22203 for r in res)      # the stand-in's number: that library really was the one bound
