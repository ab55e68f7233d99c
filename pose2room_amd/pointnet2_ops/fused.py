"""Fused vote-aggregation op (ball query + grouping + shared MLP + max-pool).

Placeholder until the gfx950 MFMA kernel lands: `available()` is False, so
PointnetSAModuleVotes runs the unfused HIP op chain.
"""


def available():
    return False


def supports(mlp_module, nsample):
    return False


def sa_votes(xyz, new_xyz, features, radius, nsample, mlp_module):
    raise RuntimeError("fused sa_votes kernel is not built")
