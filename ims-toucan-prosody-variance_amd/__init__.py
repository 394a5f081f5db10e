"""MI355X-native ToucanTTS inference path (acoustic model + vocoder).

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
all arithmetic runs in hand-written HIP kernels behind the C ABI declared in
``include/toucan_tts.h`` (``csrc/`` -> ``libtoucan_hip.so``).
"""
__version__ = "0.1.0"
