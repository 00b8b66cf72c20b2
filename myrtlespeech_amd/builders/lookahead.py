"""Mirror of myrtlespeech/builders/lookahead.py:5-31."""
from myrtlespeech_amd.model.lookahead import Lookahead


def build(lookahead_cfg, input_features: int) -> Lookahead:
    return Lookahead(in_features=input_features, context=lookahead_cfg.context)
