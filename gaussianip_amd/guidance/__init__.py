"""AHDS / ANPG guidance step (the SD1.5 + ControlNet half of GaussianIP's hot path)."""
from .ahds import AHDSSchedule, optimized_dual_gaussian, timestep_table  # noqa: F401
from .ipa_guidance import GuidanceConfig, PromptEmbeddings, StableDiffusionGuidance  # noqa: F401
from .refine import ViewConsistentRefiner, ddim_step, refine_timesteps  # noqa: F401
