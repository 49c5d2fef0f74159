"""MI355X-native implementation of the GAN+VAE training hot path of kartikkadur/MasterThesis
(AdaINModel / BaseModel driven by train.py).  Python host code mirrors the reference's
``src/`` surface; all arithmetic runs in ``libmt_hip.so`` (hand-written gfx950 HIP kernels).
"""
__version__ = "0.1.0"
