"""BaseModel -- same step, generator chosen by flags: style encoder reparameterised (``--reparam``) or
plain; decoder ``DecoderConcat`` (``--concat``) or ``Decoder`` (reference src/models/base_model.py:9-96)."""
from .core import networks
from .translation import TranslationModel


class BaseModel(TranslationModel):
    def __init__(self, args):
        super().__init__(args)
        self.latent_dim = args.latent_dim
        self.reparam = bool(args.reparam)
        self.model.content_encoder = networks.ContentEncoder(args.input_dim, dim=args.dim, norm_layer=args.enc_norm)
        if args.reparam:
            self.model.style_encoder = networks.ReparameterizedStyleEncoder(
                args.input_dim, output_dim=self.latent_dim, dim=args.dim, num_domains=args.num_domains,
                norm_layer=None, activation="lrelu")
        else:
            self.model.style_encoder = networks.StyleEncoder(
                args.input_dim, output_dim=self.latent_dim, dim=args.dim, num_domains=args.num_domains,
                activation="lrelu")
        dec_cls = networks.DecoderConcat if args.concat else networks.Decoder
        self.model.decoder = dec_cls(args.input_dim, dim=self.model.content_encoder.output_dim,
                                     num_domains=args.num_domains, latent_dim=self.latent_dim, up_type=args.up_type,
                                     norm_layer=args.dec_norm, dropout=args.use_dropout)
        self._build_training_side(args)
