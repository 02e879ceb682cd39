"""Name -> class registries with detectron2's `Registry` surface (`register()` as decorator or
call, `get(name)`), so the reference's three plugin points resolve unchanged:
  META_ARCH_REGISTRY            'Distillator' + cfg.MODEL.META_ARCHITECTURE        [ref: train.py:247-248,262]
  CUSTOMIZED_DETECTORS_REGISTRY cfg.MODEL.DISTILLATOR.{STUDENT,TEACHER}.META_ARCH   [ref: models/customized_detectors/build.py:11-17]
  ADAPTERS_REGISTRY             cfg.MODEL.DISTILLATOR.ADAPTER.META_ARCH             [ref: models/adapters/build.py:10-17]
"""


class Registry:
    def __init__(self, name):
        self._name = name
        self._map = {}

    def register(self, obj=None):
        if obj is None:
            return lambda o: self.register(o)
        name = obj.__name__
        if name in self._map:
            raise KeyError("'%s' is already registered in %s" % (name, self._name))
        self._map[name] = obj
        return obj

    def get(self, name):
        if name not in self._map:
            raise KeyError("no object named '%s' in the %s registry (have: %s)" % (name, self._name, sorted(self._map)))
        return self._map[name]

    def __contains__(self, name):
        return name in self._map

    def __iter__(self):
        return iter(self._map.items())


META_ARCH_REGISTRY = Registry("META_ARCH")
CUSTOMIZED_DETECTORS_REGISTRY = Registry("CUSTOMIZED_DETECTORS")
ADAPTERS_REGISTRY = Registry("ADAPTERS")
