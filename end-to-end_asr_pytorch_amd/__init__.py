"""end-to-end_asr_pytorch_amd — MI355X-native Speech-Transformer / CTC / CIF forward+loss path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic on the path runs in
hand-written gfx950 HIP kernels behind the C-ABI of include/asr_hip.h (csrc/libasr_hip.so, loaded with ctypes).
Import as `asr_amd` (see asr_amd.py at the repo root - the directory name is not a Python identifier).
"""
from . import _lib, checkpoint, data, decode, mask_lm, modules, ops, trainer, utils  # noqa: F401
from .decode import GreedyDecoder, ctc_greedy_decode  # noqa: F401
from .trainer import Trainer  # noqa: F401
from ._lib import build_library, lib  # noqa: F401
from .ctc_model import CTC_Model  # noqa: F401
from .loss import cal_ce_loss, cal_ctc_ce_loss, cal_ctc_qua_ce_loss, cal_loss, ctc_loss  # noqa: F401
from .modules import (Attention_Assigner, CIF_Model, Conv1d, Conv2dSubsample, Conv_CTC_Transformer, CTC_Transformer,  # noqa: F401
                      Decoder, Decoder_CIF, DecoderLayer, Encoder, EncoderLayer, MultiheadAttention, PositionalEncoding,
                      PositionwiseFeedForward, Transformer, dropout_site_keys, dropout_thr16, get_precision, manual_seed, precision,
                      set_precision)
