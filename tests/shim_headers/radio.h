// compile-check stand-in for the reference's GUI class (see README.md in this directory)
#pragma once
#include <QObject>
#include "fm-processor.h"
class RadioInterface : public QObject {
    Q_OBJECT
public slots:
    void showMetaData(const fmProcessor::SMetaData *) {}
};
