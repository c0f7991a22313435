"""`vita` import surface of the reference (VITA-MLLM/VITA) served by vita_amd (MI355X HIP path)."""
