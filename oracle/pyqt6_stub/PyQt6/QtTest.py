from ._dummy import (Dummy, QDir, QObject, QSettings, module_getattr, pyqtSignal, pyqtSlot)

__getattr__ = module_getattr({})
