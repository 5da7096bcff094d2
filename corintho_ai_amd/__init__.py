"""corintho_ai_amd -- MI355X-native self-play engine for Corintho (hot path of
maxjiang216/corintho-ai behind the reference's Trainer interface)."""
from .trainer import (NET_MLP12X100, NET_MLP12X100_H3, NET_MLP12X100_X3, NET_MLP12X100_X6, NET_RESCNN4,  # noqa: F401
                      NET_RESCNN4_H3, NET_RESCNN4_X3, NET_RESCNN4_X6, Trainer, expand_samples)
from .tourney import Tourney  # noqa: F401
