"""Mirror of the inference side of `puresound.nnet` (see the module docstrings for file:line parity)."""
from types import SimpleNamespace

from .base_nn import SiMoTaskWrapModule, SoTaskWrapModule
from .loss.sdr import SDRLoss
from .conv_tasnet import TCN, ConvTasNet, GatedTCN
from .dprnn import DPRNN
from .lobe.encoder import ConvEncDec, FbankEnc, FreeEncDec
from .lobe.pooling import AttentiveStatisticsPooling
from .lobe.trivial import FiLM, Gate, Magnitude, SpecAugment
from .skim import MemLSTM, SegLSTM, SkiM
from .unet import Unet, UnetTcn
from .dpcrn import DPCRN, DPRNNblock2D
from .dparn import DPARN, DPARN_Mout, DPARNblock2D
from .lobe.attention import MhaSelfAttenLayer
from .lobe.rnn import SingleRNN
from .lobe.cnn import DepthwiseSeparableConv1d

# the class namespace the parity tests hand to tests/golden/cases.build()
class _Namespace(SimpleNamespace):
    def __getattr__(self, name):
        if name == "StreamingSkiM":  # lives in puresound_amd.streaming, which imports this package
            from ..streaming.skim_inference import StreamingSkiM
            return StreamingSkiM
        raise AttributeError(name)


NS = _Namespace(SoTaskWrapModule=SoTaskWrapModule, SiMoTaskWrapModule=SiMoTaskWrapModule, SDRLoss=SDRLoss, TCN=TCN, ConvTasNet=ConvTasNet, GatedTCN=GatedTCN,
                     ConvEncDec=ConvEncDec, FreeEncDec=FreeEncDec,
                     AttentiveStatisticsPooling=AttentiveStatisticsPooling, DPRNN=DPRNN, SkiM=SkiM, MemLSTM=MemLSTM, Unet=Unet, UnetTcn=UnetTcn,
                DPCRN=DPCRN, DPRNNblock2D=DPRNNblock2D, DPARN=DPARN, DPARN_Mout=DPARN_Mout, DPARNblock2D=DPARNblock2D,
                MhaSelfAttenLayer=MhaSelfAttenLayer, SingleRNN=SingleRNN, Magnitude=Magnitude, SpecAugment=SpecAugment, FbankEnc=FbankEnc, SegLSTM=SegLSTM, FiLM=FiLM, Gate=Gate, DepthwiseSeparableConv1d=DepthwiseSeparableConv1d)
