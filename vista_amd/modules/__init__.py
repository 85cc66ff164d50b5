from .encoders.modules import GeneralConditioner  # noqa: F401  (configs/inference/vista.yaml: `target: vwm.modules.GeneralConditioner`)
