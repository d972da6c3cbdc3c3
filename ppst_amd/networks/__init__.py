"""Network registry with the reference's lookup rule (models/networks/__init__.py:6-49):
``create_network(opt, name, mode)`` returns the class ``<name><mode>`` found
case-insensitively (underscores ignored) in module ``<mode>``; it must subclass
BaseNetwork and take ``opt``."""
import importlib

from .base_network import BaseNetwork


def find_network_using_name(target_network_name, filename):
    target = (target_network_name + filename).replace("_", "").lower()
    mod = importlib.import_module("ppst_amd.networks." + filename)
    cls = None
    for name, obj in vars(mod).items():
        if name.lower() == target:
            cls = obj
    assert cls is not None, "In %s, there should be a class whose name matches %s in lowercase without underscore(_)" % (
        filename, target)
    assert issubclass(cls, BaseNetwork), "Class %s should be a subclass of BaseNetwork" % cls
    return cls


def create_network(opt, network_name, mode, verbose=True):
    if network_name is None:
        return None
    net = find_network_using_name(network_name, mode)(opt)
    if verbose and getattr(opt, "local_rank", 1) == 0:
        net.print_architecture(verbose=True)
    return net
