"""MI355X-native SSV filter behind HAVAC's host API.

Nothing is imported here: every module loads the native libraries it needs when it is first used, and fails loudly
if they are missing (there is no CPU fallback).

    hw_client   HavacHwClient, the mirror of the reference's host/HavacHwClient (C ABI: include/havac_dev.h)
    havac       Havac / HavacHit / HavacWindow, the mirror of host/Havac.hpp (C++ class in csrc/host, libhavac.so),
                and the host-only stages: pack_fasta, text_and_patches, project_hmm, resolve_hits, merge_windows
    ssv         SsvContext: the kernel ABI on caller-owned device memory and HIP stream
    dist        gather_hits, ShardedSsv: one process per GPU, column shards, RCCL gather of the records
    synth       synthetic inputs for tests and bench.py
"""
