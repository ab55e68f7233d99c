"""Host-side mirror of the reference's `models/p2rnet` (+ models/loss.py,
models/training.py) for the P2RNet hot path: same registry names
(`METHODS['P2RNet']`, `MODULES[{'STGCN','CenterVoteModule','ProposalNet'}]`,
`LOSSES[{'BoxNetDetectionLoss','Null'}]`), constructor signature `(cfg, optim_spec)`,
`cfg.config[...]` keys and `state_dict` layout, so reference checkpoints load and
the train_step / test_step loop drops in."""
from .registers import METHODS, MODULES, LOSSES  # noqa: F401
from . import modules  # noqa: F401  (registers the classes)
from . import loss  # noqa: F401
from .config import P2RConfig, default_config  # noqa: F401
