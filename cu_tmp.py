import time, torch
from simmr_amd import Engine, CustomShortErrorProfile
from tests import _model
e = Engine(0)
e.stage_synthetic(0, [100_000_000], 2)
prof = CustomShortErrorProfile(_model.synthetic_short_model()).pod()
from simmr_amd.engine import Reads
for n in (20_000_000,):
    info = e.pe_plan(0, prof, n, 42)
    out = Reads.allocate(info.n_reads, info.total_bases, e.device, 33)
    e.pe_emit(0, out)
    print(n, "reads: plan %.2f ms emit %.2f ms -> %.3e reads/s (emit only)" % (e.last_plan_ms(), e.last_emit_kernel_ms(), n / e.last_emit_kernel_ms() * 1e3), "bases", info.total_bases)
