"""vita/model/__init__.py of the reference: the model class."""
from vita_amd.model import VITAMixtralForCausalLM  # noqa: F401
