"""MI355X-native MFVI deep-image-prior hot path (drop-in for the MeanFieldVI path of Cardio-AI/mfvi-dip-mia).

The directory name carries a hyphen (the project name); import it as `mfvi_dip_mia_amd`
(the shim package next to it points its __path__ here)."""
