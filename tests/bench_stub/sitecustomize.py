"""TEST INFRASTRUCTURE ONLY (tests/test_bench_launch.py puts this directory on PYTHONPATH).

Lets `python3 bench.py --gpus N --backend gloo` run its whole control flow on a box without a GPU:
`simmr_amd.engine.Engine` is replaced by a stand-in that computes the shard with the CPU oracle, and the two
torch.cuda calls bench.py makes outside the engine become no-ops.  bench.py itself carries no test hook —
on a GPU box, without this directory on the path, it can only ever reach the HIP library."""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def _install():
    import numpy as np
    import torch

    import simmr_amd.engine as real
    from simmr_amd import _abi
    from tests import _oracle, _synth

    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None

    class Info:
        def __init__(self, n_reads, total_bases):
            self.n_reads, self.total_bases, self.slot_bytes = n_reads, total_bases, 0

    class OracleEngine:
        """the calls bench.py makes on an Engine, answered by oracle/liboracle.so"""

        def __init__(self, device=0):
            self.lib = _oracle.load()
            self.genomes = {}
            self.cnt = np.zeros(_abi.N_COUNTERS, dtype=np.int64)
            self.shard = None
            self.ms = [0.0, 0.0]

        device = property(lambda self: torch.device("cpu"))

        def stage_synthetic(self, idx, lens, seed):
            self.genomes[idx] = _oracle.HostGenome(_synth.synthetic_contigs(list(lens), seed))

        def set_read_slots(self, slot_bytes):
            if slot_bytes:
                raise RuntimeError("the stand-in engine writes the compact layout only: --layout compact")

        def outer_summarize(self, idx, seed, slot_first, slot_count):
            from tests.test_multi_rank_cpu import outer_accept_bits, replay_outer
            acc = outer_accept_bits(self.lib, len(self.genomes[idx].contigs), seed, slot_first + slot_count)
            (u0, e0), (u1, e1) = (replay_outer(acc, slot_first, slot_first + slot_count, s) for s in (0, 1))
            return u0, u1, e0, e1

        def pe_plan(self, idx, prof, genome_reads, seed, first=0, count=_abi.U64_MAX, start=(0, 0)):
            if start != (0, 0):  # the claimed position of the outer stream must be a real one at or before the shard
                from tests.test_multi_rank_cpu import outer_accept_bits, replay_outer
                acc = outer_accept_bits(self.lib, len(self.genomes[idx].contigs), seed, start[0])
                assert replay_outer(acc, 0, start[0], 0) == (start[1], 0) and start[1] <= first, start
            t = time.perf_counter()
            self.shard = _oracle.simulate_pe(self.lib, self.genomes[idx], prof, genome_reads, seed, first, count, qual_offset=33)
            self.ms = [0.0, (time.perf_counter() - t) * 1e3]
            return Info(self.shard.n_reads, self.shard.total_bases)

        def long_plan(self, idxs, genome_reads, prof, seed, first=0, count=_abi.U64_MAX):
            t = time.perf_counter()
            self.shard = _oracle.simulate_long(self.lib, [self.genomes[i] for i in idxs], genome_reads, prof, seed, first, count,
                                               qual_offset=33)
            self.ms = [0.0, (time.perf_counter() - t) * 1e3]
            return Info(self.shard.n_reads, self.shard.total_bases)

        def pe_emit(self, read_id_base, out):
            h = self.shard.trimmed()
            n, tb = self.shard.n_reads, self.shard.total_bases
            out.seq[:tb] = torch.from_numpy(h["seq"].copy())
            out.qual[:tb] = torch.from_numpy(h["qual"].copy())
            out.seq_off[:n + 1] = torch.from_numpy(h["seq_off"].astype(np.int64))
            out.start[:n] = torch.from_numpy(h["start"].astype(np.int64))
            out.end[:n] = torch.from_numpy(h["end"].astype(np.int64))
            acgt = int(np.isin(h["seq"], np.frombuffer(b"ACGT", np.uint8)).sum())
            self.cnt[_abi.CNT_READS] += n
            self.cnt[_abi.CNT_BASES] += tb
            self.cnt[_abi.CNT_ACGT_BASES] += acgt
            self.cnt[_abi.CNT_QUAL_SUM] += int(h["qual"].astype(np.int64).sum()) - 33 * tb

        long_emit = pe_emit

        def counters_reset(self):
            self.cnt[:] = 0

        def counters_to(self, tensor):
            tensor.copy_(torch.from_numpy(self.cnt))

        def last_emit_kernel_ms(self):
            return self.ms[1]

        def last_plan_ms(self):
            return self.ms[0]

        def emit_kernel_ms_mean(self, last_n):
            return self.ms[1]

        def set_plan_overlap(self, on):
            pass

        def close(self):
            pass

    real.Engine = OracleEngine


if os.environ.get("SIMMR_BENCH_STUB") == "1":
    _install()
