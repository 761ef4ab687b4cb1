from ._dummy import (Dummy, QByteArray, QDataStream, QDir, QObject, QSettings, module_getattr, pyqtSignal, pyqtSlot)

__getattr__ = module_getattr({})
