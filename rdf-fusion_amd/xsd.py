"""xsd:dateTime / xsd:date / xsd:time -> the device-side Timestamp (value * 10^18 on the XSD time line, has-timezone).

Host-side helper for the data generators and tests.  In the reference the host already holds this pair:
`DateTime::timestamp()` / `Date::timestamp()` / `Time::timestamp()` (lib/model/src/xsd/date_time.rs:379,631) give
`Timestamp { value: Decimal, timezone_offset: Option<TimezoneOffset> }` (:1599-1603); a Rust host copies `value`'s i128
and `timezone_offset.is_some()` into rdfgpu_typed_value (include/rdfgpu.h).  This module restates the W3C
timeOnTimeline function (https://www.w3.org/TR/xmlschema11-2/#vp-dt-timeOnTimeline; the reference's copy is
date_time.rs:2055-2079) so that Python callers can do the same from lexical forms.
"""
import re
from fractions import Fraction

SCALE = 10 ** 18          # Decimal = i128 * 10^-18 (lib/model/src/xsd/decimal.rs:9-21)


def days_in_month(year, month):
    """https://www.w3.org/TR/xmlschema11-2/#f-daysInMonth (year None = a leap year is possible)"""
    if month == 2:
        if year is None or (year % 4 == 0 and (year % 100 != 0 or year % 400 == 0)):
            return 29
        return 28
    return 30 if month in (4, 6, 9, 11) else 31


def time_on_timeline(year=None, month=None, day=None, hour=None, minute=None, second=None, tz_minutes=None):
    """Seconds on the time line (an exact Fraction) of a seven-property model; absent properties as in the spec."""
    yr = 1971 if year is None else year - 1
    mo = 12 if month is None else month
    da = days_in_month(yr + 1, mo) - 1 if day is None else day - 1
    hr = hour or 0
    mi = (minute or 0) - (tz_minutes or 0)
    se = Fraction(second or 0)
    return (31_536_000 * yr + 86400 * (yr // 400 - yr // 100 + yr // 4)
            + 86400 * sum(days_in_month(yr + 1, m) for m in range(1, mo)) + 86400 * da + 3600 * hr + 60 * mi + se)


def _scaled(seconds):
    v = seconds * SCALE
    if v.denominator != 1:
        raise ValueError("more than 18 fractional digits")
    v = int(v)
    if not -(1 << 127) <= v < (1 << 127):
        raise OverflowError("timestamp outside the i128 Decimal range")
    return v


_TZ = r"(Z|[+-]\d\d:\d\d)?"


def _tz(text):
    if not text:
        return None
    if text == "Z":
        return 0
    sign = -1 if text[0] == "-" else 1
    return sign * (int(text[1:3]) * 60 + int(text[4:6]))


def parse_date_time(lexical):
    """'2002-04-02T12:00:00-05:00' -> (value * 10^18, has_timezone); 24:00:00 is the next day's 00:00:00"""
    m = re.fullmatch(r"(-?\d{4,})-(\d\d)-(\d\d)T(\d\d):(\d\d):(\d\d(?:\.\d+)?)" + _TZ, lexical)
    if not m:
        raise ValueError(lexical)
    tz = _tz(m.group(7))
    v = time_on_timeline(int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), Fraction(m.group(6)), tz)
    return _scaled(v), tz is not None


def parse_date(lexical):
    m = re.fullmatch(r"(-?\d{4,})-(\d\d)-(\d\d)" + _TZ, lexical)
    if not m:
        raise ValueError(lexical)
    tz = _tz(m.group(4))
    return _scaled(time_on_timeline(int(m.group(1)), int(m.group(2)), int(m.group(3)), tz_minutes=tz)), tz is not None


def parse_time(lexical):
    """Time::from_parts (date_time.rs:347-369): 24:00:00 is 00:00:00 of the same (reference) day"""
    m = re.fullmatch(r"(\d\d):(\d\d):(\d\d(?:\.\d+)?)" + _TZ, lexical)
    if not m:
        raise ValueError(lexical)
    h, mi, s = int(m.group(1)), int(m.group(2)), Fraction(m.group(3))
    if h == 24 and mi == 0 and s == 0:
        h = 0
    tz = _tz(m.group(4))
    return _scaled(time_on_timeline(hour=h, minute=mi, second=s, tz_minutes=tz)), tz is not None
