"""``import admm_net_amd.dropin.activate`` as the first line of a script: the shims take precedence from here on."""
from . import activate as _activate

_activate()
