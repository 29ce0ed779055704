from .wavelet import wavelet_dec_2, wavelet_enc_2  # noqa: F401
from .metrics import Metric, Accuracy, Precision, Recall, F1  # noqa: F401
