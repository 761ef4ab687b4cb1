from ._dummy import (Dummy, QDir, QObject, QPainterPath, QSettings, module_getattr, pyqtSignal, pyqtSlot)

__getattr__ = module_getattr({})
