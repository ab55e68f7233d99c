from .network import P2RNet  # noqa: F401
from .stgcn import STGCN  # noqa: F401
from .vote_center import CenterVoteModule  # noqa: F401
from .proposal_net import ProposalNet  # noqa: F401
