from .wavelet import wavelet_dec_2, wavelet_enc_2  # noqa: F401
