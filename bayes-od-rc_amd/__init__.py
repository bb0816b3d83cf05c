"""bayes-od-rc_amd: MI355X-native BayesOD inference hot path.

Python mirror of the reference's call surface over the C ABI of
``libbayesod_hip.so`` (include/bayesod.h).  There is no CPU fallback: every compute entry point
raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"
