from adaface_amd.ldm.modules.subj_basis_generator import SubjBasisGenerator  # noqa: F401
