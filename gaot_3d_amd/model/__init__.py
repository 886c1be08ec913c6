"""Model factory with the reference's signature (src/model/__init__.py:8-28)."""
from .gaot_3d import GAOT3D


def init_model(input_size: int, output_size: int, model: str, config=None):
    if model.lower() == "gaot_3d":
        return GAOT3D(input_size=input_size, output_size=output_size, magno_config=config.magno,
                      attn_config=config.transformer, latent_tokens=config.latent_tokens)
    raise ValueError(f"model {model} not supported currently!")
