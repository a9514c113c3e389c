"""Process-wide switches shared by distributed.py and recipe.py (kept apart to avoid an import cycle)."""
import os

import torch.distributed as dist

FORCE_COLLECTIVES = os.environ.get("NVTE_AMD_FORCE_COLLECTIVES", "0") == "1"


def collectives_active(group=None) -> bool:
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group=group) > 1 or FORCE_COLLECTIVES
