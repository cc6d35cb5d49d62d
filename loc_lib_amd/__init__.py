"""loc_lib_amd — MI355X-native point-cloud registration hot path behind the LocUtils matcher API."""
