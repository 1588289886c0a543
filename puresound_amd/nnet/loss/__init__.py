from .sdr import SDRLoss, inactive_sdr_loss, l2_norm, si_snr  # noqa: F401
