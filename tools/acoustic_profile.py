"""Acoustic model alone (stage API, vocode=False) at the bench's configuration, a few passes - run under
`rocprofv3 --kernel-trace --stats` to see every kernel of the acoustic stream (also the row-wise ones tts_profile does not time)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import fixture_weights as fw, native, synthetic as syn

dev = torch.device("cuda:0")
pipe = native.NativePipeline(fw.acoustic_state_dict(), None, None, dev, precision=sys.argv[1] if len(sys.argv) > 1 else "bf16")
B, L = 32, 128
texts = [torch.from_numpy(syn.utterance_features(i, L, word_boundaries=False)).to(dev) for i in range(B)]
embs = torch.stack([torch.from_numpy(syn.utterance_embedding(i)) for i in range(B)]).to(dev)
durs = [torch.full((L,), 5, dtype=torch.int32, device=dev) for _ in range(B)]
zs = [torch.from_numpy(syn.postflow_noise(i, 5 * L)).to(dev) for i in range(B)]
packed = pipe.pack_inputs(texts, embs, [syn.LANG_EN] * B, durations=durs)
z_sq = pipe.squeeze_noise(zs, [5 * L] * B)
for _ in range(6):
    pipe.forward(None, None, packed=packed, z_sq=z_sq, vocode=False)
torch.cuda.synchronize()
