"""What a HIP-event pair around one kernel launch adds to the kernel's own duration (the bench brackets the
sweep that way): the same bracket around a kernel that does nothing, in the same position of the tick loop."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import build_engine  # noqa: E402

eng, info = build_engine("C3", 0, 1, torch.device("cuda", 0))
eng.run(50)
st = eng.store
a = torch.zeros(4, dtype=torch.float64, device="cuda")
y = torch.zeros(4, dtype=torch.float64, device="cuda")
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
for e0, e1 in ev:
    eng.run(1)                                   # a full tick before: the bracket sits behind a busy stream
    e0.record()
    st.ctx.check(st.lib.zrk_selftest_math(st.ctx.handle, 0, a.data_ptr(), a.data_ptr(), y.data_ptr(), 1, st._stream()), "noop")
    e1.record()
torch.cuda.synchronize()
t = np.array([e0.elapsed_time(e1) for e0, e1 in ev]) * 1e3
print(f"event pair around a one-thread kernel behind a tick: median {np.median(t):.2f} us, mean {t.mean():.2f} us, min {t.min():.2f} us")
