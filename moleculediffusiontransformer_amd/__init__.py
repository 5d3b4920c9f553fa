"""MI355X-native inverse-diffusion sampling path behind the QMDiffusion / QMDiffusionForward surface."""
