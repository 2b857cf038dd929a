#pragma once
// (Radar.cpp binds its dynamic-reconfigure callback with boost::bind; the adapter does not)
