"""vita/model/builder.py of the reference: load_pretrained_model (same signature, same 4-tuple)."""
from vita_amd.model.builder import load_pretrained_model  # noqa: F401
