"""alias package: see holoagent_amd/compat/README.md"""
