"""AdaINModel -- content encoder + reparameterised style encoder + AdaIN decoder + two (multi-scale)
discriminators (reference src/models/adain_model.py:10-81).  The step itself lives in
``translation.TranslationModel``.  Like the reference, ``--concat`` / ``--reparam`` are ignored here."""
from .core import networks
from .translation import TranslationModel


class AdaINModel(TranslationModel):
    reparam = True

    def __init__(self, args):
        super().__init__(args)
        self.latent_dim = args.latent_dim
        self.model.content_encoder = networks.ContentEncoder(args.input_dim, dim=args.dim, norm_layer=args.enc_norm)
        self.model.style_encoder = networks.ReparameterizedStyleEncoder(
            args.input_dim, output_dim=self.latent_dim, dim=args.dim, num_domains=args.num_domains, norm_layer=None,
            activation="lrelu")
        self.model.decoder = networks.AdaINDecoder(
            args.input_dim, dim=self.model.content_encoder.output_dim, num_domains=args.num_domains,
            latent_dim=self.latent_dim, up_type=args.up_type, norm_layer=args.dec_norm, dropout=args.use_dropout)
        self._build_training_side(args)
