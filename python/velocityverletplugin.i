/* SWIG interface of the Python module `velocityverletplugin` for installs that have a real OpenMM (>= 8.2, HIP platform).
   Same module name and class as the reference's python/velocityverletplugin.i, but generated from this build's own header
   instead of a hand-kept copy of the method list (so the signatures cannot drift: quirk Q7 in SURVEY.md), and with the
   unit decoration written once as a table.  Not buildable in the development image (no swig, no OpenMM). */
%module velocityverletplugin

%import(module="openmm.openmm") "swig/OpenMMSwigHeaders.i"
%include "swig/typemaps.i"
%include "std_vector.i"
%include "std_pair.i"
namespace std {
  %template(vectord) vector<double>;
  %template(vectori) vector<int>;
  %template(pairii) pair<int, int>;
  %template(vectorpairii) vector< pair<int, int> >;
}

%{
#include "OpenMM.h"
#include "OpenMMDrude.h"
#include "openmm/VVIntegrator.h"
%}

%pythoncode %{
import openmm.unit as _u
%}

/* getter -> unit it is returned in (OpenMM's MD unit system) */
%define VV_UNIT(GETTER, UNIT)
%pythonappend OpenMM::VVIntegrator::GETTER() const %{
    val = _u.Quantity(val, UNIT)
%}
%enddef
VV_UNIT(getTemperature,      _u.kelvin)
VV_UNIT(getDrudeTemperature, _u.kelvin)
VV_UNIT(getFrequency,        1 / _u.picosecond)
VV_UNIT(getDrudeFrequency,   1 / _u.picosecond)
VV_UNIT(getFriction,         1 / _u.picosecond)
VV_UNIT(getDrudeFriction,    1 / _u.picosecond)
VV_UNIT(getMaxDrudeDistance, _u.nanometer)
VV_UNIT(getMirrorLocation,   _u.nanometer)
VV_UNIT(getCosAcceleration,  _u.nanometer / _u.picosecond ** 2)
%pythonappend OpenMM::VVIntegrator::getElectricField() const %{
    val = _u.Quantity(val, _u.kilojoule / _u.nanometer / _u.elementary_charge).in_units_of(_u.volt / _u.nanometer)
%}
%pythonappend OpenMM::VVIntegrator::getViscosity() %{
    val = (_u.Quantity(val[0], _u.nanometer / _u.picosecond),
           _u.Quantity(val[1], _u.picosecond / (_u.dalton * _u.item) * _u.nanometer).in_units_of((_u.pascal * _u.second) ** -1))
%}

/* protected Integrator plumbing stays out of Python */
%ignore OpenMM::VVIntegrator::propagateNHChain;
%include "openmm/VVIntegrator.h"
