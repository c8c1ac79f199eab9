"""st_ito -- MI355X-native drop-in for the inference-time-optimisation hot path of
csteinmetz1/st-ito (the ES evaluate-population step).  Same module and function names as the
reference for that path: st_ito.utils.{load_param_model,get_param_embeds},
st_ito.style_transfer.{run_es,process_audio,load_plugins,parameters_to_dict},
st_ito.effects.Basic*.  Everything below the Python surface is hand-written HIP for gfx950
behind the C ABI in include/stito_hip.h; there is no CPU fallback."""
__version__ = "0.1.0"
