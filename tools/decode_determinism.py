"""Decode determinism screen: the same prompt prefilled and decoded repeatedly (KV cache rewound with set_kv_len) must give the
same token ids every time -- the fused-RMSNorm weight-streaming kernels use raw barriers and counted waits, and any difference
between repetitions is a synchronisation bug.   python tools/decode_determinism.py [repeats] [new_tokens]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from streammind_amd.native import NativeModel, PathConfig

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_new = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = PathConfig(llm_layers=32, max_frames_per_call=1, vit_layers=2)
model = NativeModel(cfg)
bench.random_weights_into(model, cfg, 1)
bench.random_llm_weights_into(model, cfg, 2)
model.finalize()
s = model.open_stream(max_frames=512, max_seq=2048)
g = torch.Generator(device="cuda").manual_seed(11)
s.write_tokens(0, torch.randn(300, cfg.conn_d_model, generator=g, device="cuda"))
ids = torch.cat([torch.randint(3, cfg.llm_vocab, (40,), generator=g, device="cuda", dtype=torch.int32),
                 -(torch.arange(0, 300, device="cuda", dtype=torch.int32) + 1),
                 torch.randint(3, cfg.llm_vocab, (9,), generator=g, device="cuda", dtype=torch.int32)]).contiguous()
ref = None
bad = 0
for r in range(reps):
    s.set_kv_len(0)
    s.prefill(ids)
    out = s.decode(n_new).cpu()
    if ref is None:
        ref = out
    elif not torch.equal(ref, out):
        bad += 1
        first = int((ref != out).nonzero()[0])
        print(f"repeat {r}: differs from repeat 0 at token {first}")
print(f"decode determinism: {reps} x (prefill {ids.numel()} + {n_new} greedy tokens), {bad} mismatching repeats; first ids {ref[:8].tolist()}")
sys.exit(1 if bad else 0)
