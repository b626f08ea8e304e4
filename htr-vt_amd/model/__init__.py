"""Drop-in replacement for the reference's `model` package
(/root/reference/model_v1/model/): `from model import HTR_VT` resolves here when
`htr-vt_amd/` is first on sys.path."""
